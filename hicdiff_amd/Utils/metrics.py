"""Tile-quality metrics of the reference's evaluation loop on the HIP engine.

``tile_sums``     one call of the C ABI ``hd_tile_metrics``: the SSIM map sum and the raw moments of a batch
``batch_metrics`` the per-batch values of src/Utils/stard_metrics.py:146-160 (mse, ssim, snr, pcc) on tiles mapped
                  from [-1,1] to [0,1] (inverse_data_transform('rescaled'), src/datasets/__init__.py:214-223)
``MetricLog``     the running bookkeeping of that loop (``test_result``): psnr from the running mse, means over samples

Spearman correlation (``spearmanr`` in the reference, a rank statistic computed on the host there too) is not part of
the device path.
"""
import ctypes as C
from math import log10

import torch

from .. import _lib as L


def tile_sums(pred, target, rescale=True):
    """-> (sums: float64[8] device tensor, ssim_each: float32[B] device tensor); layout in include/hicdiff_hip.h."""
    if not (pred.is_cuda and target.is_cuda):
        raise RuntimeError("hicdiff_amd metrics run on the GPU only: move the tiles to a ROCm device")
    if pred.shape != target.shape or pred.dim() != 4 or pred.shape[1] != 1 or pred.shape[2] != pred.shape[3]:
        raise ValueError(f"expected two (B,1,S,S) tensors, got {tuple(pred.shape)} and {tuple(target.shape)}")
    lib = L.load()
    p = pred.detach().to(torch.float32).contiguous()
    t = target.detach().to(torch.float32).contiguous()
    B, S = p.shape[0], p.shape[-1]
    partial = torch.empty((B, 8), dtype=torch.float64, device=p.device)
    sums = torch.empty((8,), dtype=torch.float64, device=p.device)
    each = torch.empty((B,), dtype=torch.float32, device=p.device)
    ptr = lambda x: C.c_void_p(x.data_ptr())
    rc = lib.hd_tile_metrics(ptr(p), ptr(t), B, S, 1 if rescale else 0, ptr(partial), ptr(sums), ptr(each),
                             C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream))
    if rc != 0:
        raise L.HdError(rc, "hd_tile_metrics failed (tile size must be 1..128)")
    return sums, each


def batch_metrics(pred, target):
    """Per-batch mse / ssim / snr / pcc / psnr as floats (one device->host copy of 8 doubles)."""
    sums, _ = tile_sums(pred, target, rescale=True)
    d2, ss, st, sp, stt, spp, spt, n = sums.cpu().tolist()
    mse = d2 / n
    snr = st / d2 ** 0.5 if d2 > 0 else (0.0 if st == 0 else float("inf"))       # stard_metrics.py:153-155
    cov = n * spt - sp * st
    den = ((n * spp - sp * sp) * (n * stt - st * st)) ** 0.5
    return {"mse": mse, "ssim": ss / n, "snr": snr, "pcc": cov / den if den > 0 else float("nan"),
            "psnr": 10 * log10(1 / mse) if mse > 0 else float("inf")}


class MetricLog:
    """Running values of the reference's ``test_result`` dict (stard_metrics.py:112,146-160)."""

    def __init__(self):
        self.r = {"mse": 0.0, "ssims": 0.0, "psnr": 0.0, "ssim": 0.0, "nsamples": 0, "pccs": 0.0, "pcc": 0.0, "snrs": 0.0, "snr": 0.0}

    def update(self, pred, target):
        b = pred.shape[0]
        m = batch_metrics(pred, target)
        r = self.r
        r["nsamples"] += b
        r["mse"] += m["mse"] * b
        r["ssims"] += m["ssim"] * b
        r["psnr"] = 10 * log10(1 / (r["mse"] / r["nsamples"])) if r["mse"] > 0 else float("inf")
        r["ssim"] = r["ssims"] / r["nsamples"]
        r["snrs"] += m["snr"] * b
        r["snr"] = r["snrs"]                                                   # as upstream: the running SUM (:157)
        r["pccs"] += m["pcc"] * b
        r["pcc"] = r["pccs"] / r["nsamples"]
        return m
