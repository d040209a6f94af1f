"""fp32 CPU restatement of HiCDiff's DDPM schedules, ancestral / DDIM samplers and losses
(test infrastructure).  ``model`` is any callable ``model(x, t, cond) -> eps``.
Citations are upstream ``file:line``.
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------- schedules (fp64)

def linear_betas(T: int) -> torch.Tensor:
    """src/hicdiff.py:396-403."""
    s = 1000 / T
    return torch.linspace(s * 0.0001, s * 0.02, T, dtype=torch.float64)


def cosine_betas(T: int, s: float = 0.008) -> torch.Tensor:
    """src/hicdiff.py:405-415."""
    t = torch.linspace(0, T, T + 1, dtype=torch.float64) / T
    ac = torch.cos((t + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    return torch.clip(1 - ac[1:] / ac[:-1], 0, 0.999)


def sigmoid_betas(T: int, start=-3, end=3, tau=1) -> torch.Tensor:
    """src/hicdiff.py:417-430 (v_start / v_end are fp32 scalars, as upstream)."""
    t = torch.linspace(0, T, T + 1, dtype=torch.float64) / T
    v_start = torch.tensor(start / tau).sigmoid()
    v_end = torch.tensor(end / tau).sigmoid()
    ac = (-((t * (end - start) + start) / tau).sigmoid() + v_end) / (v_end - v_start)
    ac = ac / ac[0]
    return torch.clip(1 - ac[1:] / ac[:-1], 0, 0.999)


SCHEDULES = {"linear": linear_betas, "cosine": cosine_betas, "sigmoid": sigmoid_betas}

BUFFER_NAMES = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
    "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
    "posterior_mean_coef1", "posterior_mean_coef2", "p2_loss_weight",
)


def diffusion_buffers(schedule: str, T: int, p2_gamma: float = 0.0, p2_k: float = 1.0) -> dict:
    """The 13 fp32 buffers of GaussianDiffusion.__init__ (src/hicdiff.py:472-522), computed in
    fp64 then cast, plus the SR3 side table ``sqrt_alphas_cumprod_prev`` (length T+1,
    src/hicdiff_sr3.py:535-536) kept in fp64 as upstream does (it is a plain attribute there)."""
    betas = SCHEDULES[schedule](T)
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, dim=0)
    ac_prev = F.pad(ac[:-1], (1, 0), value=1.0)
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    b64 = {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": torch.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": torch.log(1.0 - ac),
        "sqrt_recip_alphas_cumprod": torch.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": torch.sqrt(1.0 / ac - 1),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": torch.log(post_var.clamp(min=1e-20)),
        "posterior_mean_coef1": betas * torch.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac),
        "p2_loss_weight": (p2_k + ac / (1 - ac)) ** -p2_gamma,
    }
    out = {k: v.to(torch.float32) for k, v in b64.items()}
    out["sqrt_alphas_cumprod_prev"] = torch.sqrt(F.pad(ac_prev, (1, 0), value=1.0))
    return out


def _col(v: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """extract(), src/hicdiff.py:391-394."""
    return v.gather(-1, t).reshape(-1, 1, 1, 1)


# ---------------------------------------------------------------- noise plumbing

class TorchNoise:
    """Replays the reference's RNG call order on the torch CPU global generator:
    ``torch.randn(shape)`` once, then ``torch.randn_like`` per step (src/hicdiff.py:607,599)."""

    def __init__(self, seed: int):
        self.gen = torch.Generator(device="cpu")
        self.gen.manual_seed(seed)

    def randn(self, shape):
        return torch.randn(tuple(shape), generator=self.gen, dtype=torch.float32)


# ---------------------------------------------------------------- DDPM sampler

class DiffusionRef:
    """kind: 'uncond' (src/hicdiff.py), 'cond' (src/hicdiff_condition.py), 'sr3' (src/hicdiff_sr3.py)."""

    def __init__(self, model: Callable, *, image_size: int, timesteps: int = 1000, beta_schedule: str = "sigmoid",
                 loss_type: str = "l1", kind: str = "uncond", sampling_timesteps: Optional[int] = None,
                 ddim_sampling_eta: float = 0.0, channels: int = 1, objective: str = "pred_noise"):
        assert kind in ("uncond", "cond", "sr3")
        assert objective in ("pred_noise", "pred_x0", "pred_v")        # src/hicdiff.py:461
        self.model, self.kind, self.objective = model, kind, objective
        self.image_size, self.T, self.loss_type, self.channels = image_size, timesteps, loss_type, channels
        self.buf = diffusion_buffers(beta_schedule, timesteps)
        self.sampling_timesteps = sampling_timesteps or timesteps
        self.eta = ddim_sampling_eta

    # -- one reverse step -------------------------------------------------------
    def predict_x0(self, x, t_idx: int, eps):
        """predict_start_from_noise src/hicdiff.py:529-533."""
        b = self.buf
        return b["sqrt_recip_alphas_cumprod"][t_idx] * x - b["sqrt_recipm1_alphas_cumprod"][t_idx] * eps

    def eps_at(self, x, t_idx: int, cond):
        n = x.shape[0]
        if self.kind == "sr3":
            # src/hicdiff_sr3.py:634-637: conditioning value is sqrt_alphas_cumprod_prev[t+1] as fp32 (B,1)
            level = torch.FloatTensor([self.buf["sqrt_alphas_cumprod_prev"][t_idx + 1]]).repeat(n, 1)
            return self.model(x, level, cond)
        return self.model(x, torch.full((n,), t_idx, dtype=torch.long), cond)

    def x0_from_output(self, x, t_idx: int, out):
        """model_predictions, src/hicdiff.py:562-582: what the network predicts is the noise, x0 itself, or v
        (predict_start_from_v, :548-552: x0 = sqrt_ac[t] x - sqrt(1 - ac[t]) v)."""
        b = self.buf
        if self.objective == "pred_noise":
            return self.predict_x0(x, t_idx, out)
        if self.objective == "pred_x0":
            return out
        return b["sqrt_alphas_cumprod"][t_idx] * x - b["sqrt_one_minus_alphas_cumprod"][t_idx] * out

    def p_sample(self, x, t_idx: int, cond, noise):
        """p_mean_variance + p_sample, src/hicdiff.py:584-601: clamp x0 to [-1,1], posterior mean,
        x_{t-1} = mean + exp(0.5 logvar) * z (z = 0 at t = 0)."""
        b = self.buf
        eps = self.eps_at(x, t_idx, cond)                      # (the network's output: the noise only under objective 'pred_noise')
        x0 = self.x0_from_output(x, t_idx, eps).clamp(-1.0, 1.0)
        mean = b["posterior_mean_coef1"][t_idx] * x0 + b["posterior_mean_coef2"][t_idx] * x
        if t_idx > 0:
            return mean + (0.5 * b["posterior_log_variance_clipped"][t_idx]).exp() * noise, x0, eps
        return mean, x0, eps

    def p_sample_loop(self, shape_or_cond, noise: TorchNoise, keep_every: int = 0):
        """src/hicdiff.py:603-620 / src/hicdiff_condition.py:600-623."""
        if self.kind == "uncond":
            cond, shape = None, tuple(shape_or_cond)
        else:
            cond, shape = shape_or_cond, tuple(shape_or_cond.shape)
        img = noise.randn(shape)
        kept = {self.T: img.clone()} if keep_every else None
        for t in reversed(range(self.T)):
            z = noise.randn(shape) if t > 0 else None
            img, _, _ = self.p_sample(img, t, cond, z)
            if keep_every and t % keep_every == 0:
                kept[t] = img.clone()
        return (img, kept) if keep_every else img

    def interpolate(self, x1, x2, noise: TorchNoise, t: Optional[int] = None, lam: float = 0.5):
        """src/hicdiff.py:673-691: both tiles diffused to step t (two q_sample draws, x1 first), mixed, then p_sample for i = t-1 .. 0."""
        t = self.T - 1 if t is None else t
        tb = torch.full((x1.shape[0],), t, dtype=torch.long)
        xt1 = self.q_sample(x1, tb, noise.randn(x1.shape))
        xt2 = self.q_sample(x2, tb, noise.randn(x2.shape))
        img = (1 - lam) * xt1 + lam * xt2
        for i in reversed(range(0, t)):
            z = noise.randn(img.shape) if i > 0 else None
            img, _, _ = self.p_sample(img, i, None, z)
        return img

    def ddim_sample(self, shape, noise: TorchNoise):
        """src/hicdiff.py:622-664."""
        T, S, eta = self.T, self.sampling_timesteps, self.eta
        times = list(reversed(torch.linspace(-1, T - 1, steps=S + 1).int().tolist()))
        ac = self.buf["alphas_cumprod"]
        img = noise.randn(shape)
        for time, time_next in zip(times[:-1], times[1:]):
            eps = self.eps_at(img, time, None)
            x0 = self.x0_from_output(img, time, eps).clamp(-1.0, 1.0)
            if self.objective != "pred_noise":                 # predict_noise_from_start on the clipped x0 (src/hicdiff.py:571-580,535-539)
                b = self.buf
                eps = (b["sqrt_recip_alphas_cumprod"][time] * img - x0) / b["sqrt_recipm1_alphas_cumprod"][time]
            if time_next < 0:
                img = x0
                continue
            a, an = ac[time], ac[time_next]
            sigma = eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
            c = (1 - an - sigma ** 2).sqrt()
            img = x0 * an.sqrt() + c * eps + sigma * noise.randn(shape)
        return img

    # -- training objective (forward value only) --------------------------------
    def q_sample(self, x0, t, eps):
        """src/hicdiff.py:694-700."""
        b = self.buf
        return _col(b["sqrt_alphas_cumprod"], t) * x0 + _col(b["sqrt_one_minus_alphas_cumprod"], t) * eps

    def p_losses(self, x0, t, eps, cond=None):
        """src/hicdiff.py:711-747 / src/hicdiff_condition.py:715-746: per-sample mean of l1|l2,
        times p2 weight (== 1), mean over batch."""
        x = self.q_sample(x0, t, eps)
        out = self.model(x, t, cond)
        b = self.buf
        target = eps                                           # src/hicdiff.py:733-741
        if self.objective == "pred_x0":
            target = x0
        elif self.objective == "pred_v":                       # predict_v, :542-546
            target = _col(b["sqrt_alphas_cumprod"], t) * eps - _col(b["sqrt_one_minus_alphas_cumprod"], t) * x0
        per = (out - target).abs() if self.loss_type == "l1" else (out - target) ** 2
        per = per.reshape(per.shape[0], -1).mean(dim=1) * self.buf["p2_loss_weight"].gather(-1, t)
        return per.mean()

    def p_losses_sr3(self, x0, level, eps, cond):
        """src/hicdiff_sr3.py:750-792: continuous noise level, plain mean reduction."""
        lv = level.reshape(-1, 1, 1, 1)
        x = lv * x0 + (1 - lv ** 2).sqrt() * eps
        out = self.model(x, level.reshape(-1, 1), cond)
        return ((out - eps).abs() if self.loss_type == "l1" else (out - eps) ** 2).mean()

    def sr3_draw_level(self, rng: np.random.RandomState, batch: int):
        """t ~ U{1..T}; level ~ U[sqrt_ac_prev[t-1], sqrt_ac_prev[t]] (src/hicdiff_sr3.py:754-761)."""
        t = rng.randint(1, self.T + 1)
        tab = self.buf["sqrt_alphas_cumprod_prev"]
        return torch.FloatTensor(rng.uniform(tab[t - 1], tab[t], size=batch))
