"""``MakeFunc(deg, ...)`` factory (src/functions/H_func.py:4-67); HiCDiff only ever asks for 'deno'."""
from __future__ import annotations

from .svd_replacement import Denoising


def MakeFunc(deg='deno', image_channel=1, image_size=64, device=None):
    if deg == 'deno':
        return Denoising(image_channel, image_size, device)
    raise NotImplementedError(
        f"degradation '{deg}' is outside the HiCDiff hot path (the reference hard-codes deg='deno', inference.py:44)")
