"""DDRM sampler: drop-in for ``src/functions/denoising.py``.  With the identity degradation HiCDiff selects (``Denoising``)
each step is ONE ``hd_ddrm_step`` call (epsilon-network + three-case update fused); with any other operator of
``svd_replacement`` a step is the epsilon-network, ``V^T`` of its two outputs, the three-case update in the spectral domain
(``hd_ddrm_general_update``) and ``V`` of the result."""
from __future__ import annotations

import contextlib

import torch

from .. import _lib as L
from .svd_replacement import Denoising


def compute_alpha(beta, t):
    """alpha_bar at t (src/functions/denoising.py:6-9), evaluated where ``beta`` lives."""
    beta = torch.cat([torch.zeros(1).to(beta.device), beta], dim=0)
    return (1 - beta).cumprod(dim=0).index_select(0, t + 1).view(-1, 1, 1, 1)


def efficient_generalized_steps(x, seq, model, b, H_funcs, y_0, sigma_0, etaB, etaA, etaC, cls_fn=None, classes=None,
                                device=None, noise=None, keep="all", seed=1234, tile_offset=0, skip_inert_steps=False):
    """Returns (xs, x0_preds) like the reference (lists over steps; ``keep='last'`` keeps only the
    final entries).  ``model`` is the epsilon-network module (``diffusion.model``, inference.py:109).
    ``noise``: None -> device Philox; or an object with ``randn(shape)`` replaying the reference's three
    draws per step (:92 full, :96 selected pixels, :100 full).
    ``skip_inert_steps`` (identity degradation only, off by default): with etaB == 1, while sigma_next > sigma_0 every pixel takes the
    update's third case, x_next = sqrt(a_next) (y + sqrt(sigma_next^2 - sigma_0^2) z) (:99-100) -- the network's output is multiplied by
    (1 - etaB) = 0.  Such steps then run the update kernel alone (hd_ddrm_coef.skip_network): the states are bit-identical to the full
    run's, but the x0 estimates of those steps do not exist, so ``keep='all'`` is refused.  With inference.py's defaults (50 of 1000
    steps, sigma_0 = 0.1) 47 of the 50 network evaluations are inert."""
    if cls_fn is not None:
        raise NotImplementedError("classifier guidance is not used by HiCDiff")
    if skip_inert_steps and (keep == "all" or not isinstance(H_funcs, Denoising)):
        raise ValueError("skip_inert_steps needs the identity degradation and keep='last': the skipped steps produce no x0 estimate")
    if not isinstance(H_funcs, Denoising):
        return _general_steps(x, seq, model, b, H_funcs, y_0, sigma_0, etaB, etaA, etaC, noise, keep, seed, tile_offset)
    with torch.no_grad():
        x = x.contiguous().float()
        n, c, hh, ww = x.shape
        d = c * hh * ww
        seq = list(seq)
        y = y_0.reshape(n, -1).contiguous().float()
        # fp32 alpha_bar table on the host, same arithmetic as compute_alpha on a CPU tensor
        ab = (1 - torch.cat([torch.zeros(1), b.detach().float().cpu()], dim=0)).cumprod(dim=0)
        a_T = ab[seq[-1] + 1]
        sig_T = (1 - a_T).sqrt() / a_T.sqrt()
        large = bool(sig_T > sigma_0)                       # singulars are all 1 (:26)
        inv_sing = sigma_0 if large else 0.0
        remaining = float((sig_T ** 2 - inv_sing ** 2).clamp_min(0.0).sqrt())
        init = (y.reshape(x.shape) if large else torch.zeros_like(x)) + remaining * x
        xt = (init / float(sig_T)).contiguous()            # :24-41 (V = I)
        eng = model.engine(x.device)
        seq_next = [-1] + seq[:-1]
        xs, x0_preds = [xt.clone()], []
        x0 = torch.empty_like(xt)
        # a chain bracket (Engine.chain) when nothing has to see the state between steps on this stream: device noise, last state only
        bracket = eng.chain(n, hh) if (noise is None and keep != "all") else contextlib.nullcontext()
        with bracket:
            for k, (i, j) in enumerate(zip(reversed(seq), reversed(seq_next))):
                at, at_next = ab[i + 1], ab[j + 1]
                co = L.HdDdrmCoef()
                co.sqrt_at, co.sqrt_1m_at, co.sqrt_at_next = float(at.sqrt()), float((1 - at).sqrt()), float(at_next.sqrt())
                sigma_next = (1 - at_next).sqrt() / at_next.sqrt()
                co.sigma_next, co.sigma_0, co.etaA, co.etaB, co.etaC = float(sigma_next), float(sigma_0), float(etaA), float(etaB), float(etaC)
                co.time_value = float(i)
                co.skip_network = 1 if (skip_inert_steps and float(etaB) == 1.0 and co.sigma_next > co.sigma_0) else 0
                z = None
                if noise is not None:
                    after = bool(sigma_next < sigma_0)
                    z = torch.empty((3, n, d), device=x.device, dtype=torch.float32)
                    z[0] = noise.randn((n, d))
                    za = noise.randn((n, d if after else 0))
                    if after:
                        z[1] = za
                    z[2] = noise.randn((n, d))
                eng.ddrm_step(xt, y, z, co, x0, seed=seed, tile_offset=tile_offset, step=k)
                if keep == "all":
                    xs.append(xt.clone()); x0_preds.append(x0.clone())
        if keep != "all":
            xs, x0_preds = [xt], [x0]
    return xs, x0_preds


def _general_steps(x, seq, model, b, H, y_0, sigma_0, etaB, etaA, etaC, noise, keep, seed, tile_offset):
    """src/functions/denoising.py:11-111 for an operator with its own U, singular values and V."""
    import ctypes as C
    lib = L.load()
    P = lambda t: C.c_void_p(t.data_ptr())
    with torch.no_grad():
        x = x.contiguous().float()
        dev = x.device
        n, D = x.shape[0], x.shape[1] * x.shape[2] * x.shape[3]
        if D % 4:
            raise ValueError("channels * S * S must be a multiple of 4")
        st = lambda: C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        seq = list(seq)
        singulars = H.singulars().float().contiguous()
        U_t_y = H.Ut(y_0).contiguous()
        M = U_t_y.shape[1]
        sing_m = singulars[:M].contiguous()
        ab = (1 - torch.cat([torch.zeros(1), b.detach().float().cpu()], dim=0)).cumprod(dim=0)
        # x_T (:24-44): p(x_T | x_0, y) in the spectral domain, then back to pixels
        a_T = ab[seq[-1] + 1]
        sig_T = float((1 - a_T).sqrt() / a_T.sqrt())
        large = singulars * sig_T > sigma_0
        inv = torch.zeros(D, device=dev)
        inv[:singulars.shape[0]][large] = sigma_0 / singulars[large]
        init_y = torch.zeros((n, D), device=dev)
        lidx = torch.nonzero(large).reshape(-1)
        init_y[:, lidx] = U_t_y[:, lidx] / singulars[lidx].view(1, -1)
        remaining = (sig_T ** 2 - inv.view(1, -1) ** 2).clamp_min(0.0).sqrt()
        init = (init_y + remaining * x.reshape(n, D)) / sig_T
        xt = H.V(init).view(*x.shape).contiguous()
        seq_next = [-1] + seq[:-1]
        xs, x0_preds = [xt.clone()], []
        x0 = torch.empty_like(xt)
        out = torch.empty((n, D), device=dev)
        for k, (i, j) in enumerate(zip(reversed(seq), reversed(seq_next))):
            at, at_next = ab[i + 1], ab[j + 1]
            co = L.HdDdrmCoef()
            co.sqrt_at, co.sqrt_1m_at, co.sqrt_at_next = float(at.sqrt()), float((1 - at).sqrt()), float(at_next.sqrt())
            sigma_next = (1 - at_next).sqrt() / at_next.sqrt()
            co.sigma_next, co.sigma_0, co.etaA, co.etaB, co.etaC = float(sigma_next), float(sigma_0), float(etaA), float(etaB), float(etaC)
            co.time_value = float(i)
            et = model(xt, torch.full((n,), float(i), device=dev))                       # (n, c, S, S)
            if lib.hd_ddrm_x0(P(xt), P(et), co.sqrt_at, co.sqrt_1m_at, P(x0), xt.numel(), st()) != 0:
                raise L.HdError(-1, "hd_ddrm_x0 failed")
            vt_x0, vt_et = H.Vt(x0).contiguous(), H.Vt(et).contiguous()
            z0 = z1 = z2 = None
            if noise is not None:                  # the reference's three draws (:92 full, :96 the 'after' entries only, :100 measurement-sized)
                z0 = noise.randn((n, D)).to(dev).contiguous()
                after = torch.zeros(D, dtype=torch.bool, device=dev)
                after[:M] = sing_m * float(sigma_next) < sigma_0
                aidx = torch.nonzero(after).reshape(-1)
                z1c = noise.randn((n, int(aidx.numel())))
                z1 = torch.zeros((n, D), device=dev)
                z1[:, aidx] = z1c.to(dev)
                z2 = noise.randn((n, M)).to(dev).contiguous()
            rc = lib.hd_ddrm_general_update(P(vt_x0), P(vt_et), P(U_t_y), P(sing_m), M, None if z0 is None else P(z0), None if z1 is None else P(z1),
                                            None if z2 is None else P(z2), C.byref(co), P(out), n, D, seed, tile_offset, k, st())
            if rc != 0:
                raise L.HdError(rc, "hd_ddrm_general_update failed")
            xt = H.V(out).view(*x.shape).contiguous()
            if keep == "all":
                xs.append(xt.clone()); x0_preds.append(x0.clone())
        if keep != "all":
            xs, x0_preds = [xt], [x0]
    return xs, x0_preds
