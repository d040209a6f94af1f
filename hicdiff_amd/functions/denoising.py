"""DDRM sampler as HiCDiff drives it: drop-in for ``src/functions/denoising.py`` with the identity
degradation.  Each step is one ``hd_ddrm_step`` call (epsilon-network + three-case update)."""
from __future__ import annotations

import torch

from .. import _lib as L
from .svd_replacement import Denoising


def compute_alpha(beta, t):
    """alpha_bar at t (src/functions/denoising.py:6-9), evaluated where ``beta`` lives."""
    beta = torch.cat([torch.zeros(1).to(beta.device), beta], dim=0)
    return (1 - beta).cumprod(dim=0).index_select(0, t + 1).view(-1, 1, 1, 1)


def efficient_generalized_steps(x, seq, model, b, H_funcs, y_0, sigma_0, etaB, etaA, etaC, cls_fn=None, classes=None,
                                device=None, noise=None, keep="all", seed=1234, tile_offset=0):
    """Returns (xs, x0_preds) like the reference (lists over steps; ``keep='last'`` keeps only the
    final entries).  ``model`` is the epsilon-network module (``diffusion.model``, inference.py:109).
    ``noise``: None -> device Philox; or an object with ``randn(shape)`` replaying the reference's three
    draws per step (:92 full, :96 selected pixels, :100 full)."""
    if not isinstance(H_funcs, Denoising):
        raise NotImplementedError("only the 'deno' degradation (identity H) is on the HiCDiff path")
    if cls_fn is not None:
        raise NotImplementedError("classifier guidance is not used by HiCDiff")
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
        for k, (i, j) in enumerate(zip(reversed(seq), reversed(seq_next))):
            at, at_next = ab[i + 1], ab[j + 1]
            co = L.HdDdrmCoef()
            co.sqrt_at, co.sqrt_1m_at, co.sqrt_at_next = float(at.sqrt()), float((1 - at).sqrt()), float(at_next.sqrt())
            sigma_next = (1 - at_next).sqrt() / at_next.sqrt()
            co.sigma_next, co.sigma_0, co.etaA, co.etaB, co.etaC = float(sigma_next), float(sigma_0), float(etaA), float(etaB), float(etaC)
            co.time_value = float(i)
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
