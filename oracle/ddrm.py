"""fp32 CPU restatement of the DDRM sampler as HiCDiff drives it (test infrastructure).

Only the configuration the reference ever selects is restated: degradation 'deno'
(identity H, all singular values 1; src/functions/svd_replacement.py:148-168 chosen by
src/functions/H_func.py:18-20), driven by src/Utils/metrics_diff.py:215-224 with
etaB=1.0, etaA=etaC=0.85 and the 1000-step linear beta table of :36-81,100-107.
"""
from __future__ import annotations

import numpy as np
import torch


def ddrm_betas(schedule: str = "linear", n: int = 1000) -> torch.Tensor:
    """src/Utils/metrics_diff.py:36-81 ('linear': numpy fp64 linspace cast to fp32, :108-109)."""
    if schedule == "linear":
        return torch.from_numpy(np.linspace(0.0001, 0.02, n, dtype=np.float64)).float()
    from .diffusion import sigmoid_betas
    return sigmoid_betas(n).float()


def alpha_bar(betas: torch.Tensor, t: int) -> torch.Tensor:
    """compute_alpha, src/functions/denoising.py:6-9: cumprod of (1 - [0, beta]) at index t+1 (fp32)."""
    b = torch.cat([torch.zeros(1), betas], dim=0)
    return (1 - b).cumprod(dim=0)[t + 1]


def ddrm_denoise(x, seq, model, betas, y_0, sigma_0, etaB=1.0, etaA=0.85, etaC=0.85, noise=None,
                 keep_steps=()):
    """efficient_generalized_steps (src/functions/denoising.py:11-111) specialised to identity H.

    With every singular value equal to 1 all pixels take the same branch of the three-case
    update each step.  The reference still draws three Gaussian tensors per step, in this
    order: the 'missing' one (:92), the 'after' one (:96, shape = selected pixels) and the
    'before' one (:100, full shape); ``noise.randn`` is called with exactly those shapes so a
    seeded replay consumes the generator identically.
    """
    n, c, hh, ww = x.shape
    d = c * hh * ww
    y = y_0.reshape(n, -1)
    # init x_T (:24-41)
    a_last = alpha_bar(betas, seq[-1])
    sig_T = (1 - a_last).sqrt() / a_last.sqrt()
    large = bool(sig_T > sigma_0)
    inv_sing = sigma_0 if large else 0.0
    init_y = y.clone() if large else torch.zeros_like(y)
    remaining = (sig_T ** 2 - inv_sing ** 2).clamp_min(0.0).sqrt()
    xt = ((init_y.reshape(x.shape) + remaining * x) / sig_T)
    seq_next = [-1] + list(seq[:-1])
    kept = {}
    x0_t = None
    for k, (i, j) in enumerate(zip(reversed(seq), reversed(seq_next)), start=1):
        t = torch.ones(n) * i
        at, at_next = alpha_bar(betas, i), alpha_bar(betas, j)
        et = model(xt, t)
        x0_t = (xt - et * (1 - at).sqrt()) / at.sqrt()                      # :66
        sigma_next = (1 - at_next).sqrt() / at_next.sqrt()
        v0 = x0_t.reshape(n, -1)
        before = bool(sigma_next > sigma_0)
        after = bool(sigma_next < sigma_0)
        std_c = sigma_next * etaC
        tilde_c = torch.sqrt(sigma_next ** 2 - std_c ** 2)
        std_a = sigma_next * etaA
        tilde_a = torch.sqrt(sigma_next ** 2 - std_a ** 2)
        z_missing = noise.randn((n, d))
        nxt = v0 + tilde_c * et.reshape(n, -1) + std_c * z_missing          # :92
        z_after = noise.randn((n, d if after else 0))
        if after:
            nxt = v0 + tilde_a * ((y - v0) / sigma_0) + std_a * z_after     # :95-96
        z_before = noise.randn((n, d))
        if before:
            diff_b = torch.sqrt(sigma_next ** 2 - sigma_0 ** 2 * (etaB ** 2))
            nxt = y * etaB + (1 - etaB) * v0 + diff_b * z_before            # :99-100
        xt = (at_next.sqrt() * nxt).reshape(x.shape)                        # :104
        if k in keep_steps:
            kept[k] = xt.clone()          # state after the k-th executed step
    return (xt, x0_t, kept) if keep_steps else (xt, x0_t)


class DenseH:
    """A degradation given by dense factors (test infrastructure): H = U diag(s) V^T with V (D x D), U (M x M), s (M,), all
    in the spectral ordering of the reference's operator (src/functions/svd_replacement.py: the matrices are what its
    V / U methods do to the unit vectors).  Same method names as H_functions (:3-69)."""

    def __init__(self, V, U, s):
        self.Vm, self.Um, self.s = V.float(), U.float(), s.float()

    def V(self, vec): return vec.reshape(vec.shape[0], -1) @ self.Vm.T
    def Vt(self, vec): return vec.reshape(vec.shape[0], -1) @ self.Vm
    def U(self, vec): return vec.reshape(vec.shape[0], -1) @ self.Um.T
    def Ut(self, vec): return vec.reshape(vec.shape[0], -1) @ self.Um
    def singulars(self): return self.s

    def H(self, vec):
        return self.U(self.s * self.Vt(vec)[:, :self.s.shape[0]])


def ddrm_general(x, seq, model, betas, H, y_0, sigma_0, etaB=1.0, etaA=0.85, etaC=0.85, noise=None):
    """efficient_generalized_steps (src/functions/denoising.py:11-111) for any operator H with V / Vt / Ut / singulars.
    Noise draws in the reference's order and shapes: x_T is the caller's; per step (n, D) for the default branch (:92),
    (n, #after) for the entries whose measurement is noisier than the current level (:96), (n, M) for the rest (:100)."""
    n = x.shape[0]
    D = x[0].numel()
    s = H.singulars()
    Sigma = torch.zeros(D)
    Sigma[:s.shape[0]] = s
    Uty = H.Ut(y_0)
    M = Uty.shape[1]
    # x_T (:24-44)
    a_last = alpha_bar(betas, seq[-1])
    sig_T = (1 - a_last).sqrt() / a_last.sqrt()
    large = torch.where(s * sig_T > sigma_0)[0]
    inv0 = torch.zeros(D)
    inv0[large] = sigma_0 / s[large]
    init_y = torch.zeros(n, D)
    init_y[:, large] = Uty[:, large] / s[large].view(1, -1)
    remaining = (sig_T ** 2 - inv0.view(1, -1) ** 2).clamp_min(0.0).sqrt()
    init = (init_y + remaining * x.reshape(n, D)) / sig_T
    xt = H.V(init).reshape(x.shape)
    seq_next = [-1] + list(seq[:-1])
    x0_t = None
    for i, j in zip(reversed(seq), reversed(seq_next)):
        at, at_next = alpha_bar(betas, i), alpha_bar(betas, j)
        et = model(xt, torch.ones(n) * i)
        x0_t = (xt - et * (1 - at).sqrt()) / at.sqrt()
        sigma_next = (1 - at_next).sqrt() / at_next.sqrt()
        Vx0, Vet = H.Vt(x0_t), H.Vt(et)
        SVx0 = (Vx0 * Sigma)[:, :M]
        before_l, after_l = s * sigma_next > sigma_0, s * sigma_next < sigma_0
        pad = torch.zeros(D - s.shape[0], dtype=torch.bool)
        before, after = torch.hstack((before_l, pad)), torch.hstack((after_l, pad))
        std_c = sigma_next * etaC
        til_c = torch.sqrt(sigma_next ** 2 - std_c ** 2)
        std_a = sigma_next * etaA
        til_a = torch.sqrt(sigma_next ** 2 - std_a ** 2)
        nxt = Vx0 + til_c * Vet + std_c * noise.randn((n, D))
        z_a = noise.randn((n, int(after.sum())))
        nxt[:, after] = Vx0[:, after] + til_a * ((Uty - SVx0) / sigma_0)[:, after_l] + std_a * z_a
        z_b = noise.randn((n, M))
        diff_b = torch.sqrt(sigma_next ** 2 - sigma_0 ** 2 / s[before_l] ** 2 * (etaB ** 2))
        nxt[:, before] = (Uty / s[:M])[:, before_l] * etaB + (1 - etaB) * Vx0[:, before] + diff_b * z_b[:, before_l]
        xt = (at_next.sqrt() * H.V(nxt)).reshape(x.shape)
    return xt, x0_t
