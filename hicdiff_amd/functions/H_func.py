"""``MakeFunc(deg, ...)`` factory (src/functions/H_func.py:4-67).  HiCDiff only ever asks for 'deno'; the other
degradations of the DDRM operator zoo are available too."""
from __future__ import annotations

import numpy as np
import torch

from . import svd_replacement as SV
from .svd_replacement import Denoising


def MakeFunc(deg='deno', image_channel=1, image_size=64, device=None):
    if deg == 'deno':
        return Denoising(image_channel, image_size, device)
    if deg[:2] == 'cs':
        return SV.WalshHadamardCS(image_channel, image_size, int(deg[2:]), torch.randperm(image_size ** 2, device=device), device)
    if deg[:3] == 'inp':
        if deg != 'inp_mask':
            raise ValueError("only 'inp_mask' defines the missing pixels (the reference leaves them undefined for other 'inp*' names)")
        missing = torch.randperm(image_size ** 2)[:image_size ** 2 // 2].to(device).long()
        return SV.Inpainting(image_channel, image_size, missing, device)
    if deg[:10] == 'sr_bicubic':
        factor = int(deg[10:])

        def bicubic_kernel(x, a=-0.5):
            if abs(x) <= 1:
                return (a + 2) * abs(x) ** 3 - (a + 3) * abs(x) ** 2 + 1
            if 1 < abs(x) < 2:
                return a * abs(x) ** 3 - 5 * a * abs(x) ** 2 + 8 * a * abs(x) - 4 * a
            return 0

        k = np.zeros((factor * 4))
        for i in range(factor * 4):
            k[i] = bicubic_kernel((1 / factor) * (i - np.floor(factor * 4 / 2) + 0.5))
        k = k / np.sum(k)
        kernel = torch.from_numpy(k).float()
        return SV.SRConv(kernel / kernel.sum(), image_channel, image_size, device, stride=factor)
    if deg == 'deblur_uni':
        return SV.Deblurring(torch.Tensor([1 / 9] * 9), image_channel, image_size, device)
    if deg == 'deblur_gauss':
        pdf = lambda x: torch.exp(torch.Tensor([-0.5 * (x / 10) ** 2]))
        kernel = torch.Tensor([pdf(-2), pdf(-1), pdf(0), pdf(1), pdf(2)])
        return SV.Deblurring(kernel / kernel.sum(), image_channel, image_size, device)
    if deg == 'deblur_aniso':
        pdf2 = lambda x: torch.exp(torch.Tensor([-0.5 * (x / 20) ** 2]))
        pdf1 = lambda x: torch.exp(torch.Tensor([-0.5 * (x / 1) ** 2]))
        k2 = torch.Tensor([pdf2(v) for v in range(-4, 5)])
        k1 = torch.Tensor([pdf1(v) for v in range(-4, 5)])
        return SV.Deblurring2D(k1 / k1.sum(), k2 / k2.sum(), image_channel, image_size, device)
    if deg[:2] == 'sr':
        return SV.SuperResolution(image_channel, image_size, int(deg[2:]), device)
    if deg == 'color':
        return SV.Colorization(image_size, device)
    raise ValueError(f"degradation type '{deg}' not supported")          # the reference prints an error and quits
