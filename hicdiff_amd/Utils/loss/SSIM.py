"""Mirror of the reference's ``src/Utils/loss/SSIM.py`` (DeepHiC's SSIM) on the HIP engine.

``ssim(img1, img2, window_size=11, size_average=True)`` and ``SSIM(window_size=11, size_average=True)`` keep the
reference signatures (SSIM.py:40-74).  The engine implements the configuration the reference actually uses:
one channel, window 11 (sigma 1.5), float32 device tensors of tiles up to 128x128; anything else raises -- there is
no CPU fallback on the product path.
"""
import ctypes as C

import torch

from .. import metrics as _m


def _check(img1, img2, window_size):
    if window_size != 11:
        raise NotImplementedError("the HIP SSIM kernel implements the reference's window_size=11 (sigma 1.5) only")
    if img1.shape != img2.shape or img1.dim() != 4 or img1.shape[1] != 1 or img1.shape[2] != img1.shape[3]:
        raise ValueError(f"expected two (B,1,S,S) tensors, got {tuple(img1.shape)} and {tuple(img2.shape)}")


def ssim(img1, img2, window_size=11, size_average=True):
    """SSIM.py:65-74.  Inputs are images as the reference passes them (already in [0,1])."""
    _check(img1, img2, window_size)
    sums, each = _m.tile_sums(img1, img2, rescale=False)
    if size_average:
        return (sums[1] / sums[7]).to(torch.float32)
    return each


class SSIM(torch.nn.Module):
    """SSIM.py:40-63."""

    def __init__(self, window_size=11, size_average=True):
        super().__init__()
        self.window_size = window_size
        self.size_average = size_average
        self.channel = 1

    def forward(self, img1, img2):
        return ssim(img1, img2, self.window_size, self.size_average)
