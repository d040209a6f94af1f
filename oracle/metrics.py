"""CPU restatement of the reference's tile-quality metrics (TEST INFRASTRUCTURE: imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product path).

  ssim / _ssim / create_window / gaussian   src/Utils/loss/SSIM.py:6-37,65-74   (DeepHiC's SSIM: 11x11 Gaussian
                                            window, sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2)
  batch_metrics                             src/Utils/stard_metrics.py:146-165  (per batch: mse, ssim, snr, pcc on
                                            tiles mapped to [0,1] by inverse_data_transform('rescaled', .),
                                            src/datasets/__init__.py:214-223; psnr from the running mse)
"""
from math import exp, log10

import torch
import torch.nn.functional as F


def gaussian(window_size, sigma):
    g = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return g / g.sum()


def create_window(window_size=11, channel=1):
    w1 = gaussian(window_size, 1.5).unsqueeze(1)
    w2 = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, window_size, window_size).contiguous()


def ssim(img1, img2, window_size=11, size_average=True):
    channel = img1.shape[1]
    window = create_window(window_size, channel).type_as(img1)
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, window, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=pad, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean() if size_average else ssim_map.mean(1).mean(1).mean(1)


def rescaled(x):
    return torch.clamp((x + 1.0) / 2.0, 0.0, 1.0)


def batch_metrics(pred, target):
    """pred, target: (B,1,S,S) in [-1,1].  Returns the per-batch values of stard_metrics.py:146-160."""
    out, hr = rescaled(pred), rescaled(target)
    mse = ((out - hr) ** 2).mean()
    den = ((hr - out) ** 2).sum().sqrt()
    snr = hr.sum() / den if not (den == 0 and hr.sum() == 0) else torch.tensor(0.0)
    x, y = out.flatten().double(), hr.flatten().double()
    xm, ym = x - x.mean(), y - y.mean()
    pcc = (xm * ym).sum() / (xm.pow(2).sum().sqrt() * ym.pow(2).sum().sqrt())
    return {"mse": float(mse), "ssim": float(ssim(out, hr)), "snr": float(snr), "pcc": float(pcc),
            "psnr": 10 * log10(1 / float(mse)) if float(mse) > 0 else float("inf")}
