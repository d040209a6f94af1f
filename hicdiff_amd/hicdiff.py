"""Unconditional HiCDiff: drop-in for ``src/hicdiff.py`` (Unet :255-387, GaussianDiffusion :432-755)."""
from __future__ import annotations

import torch

from ._diffusion import (DiffusionCore, HostReplayNoise, ModelPrediction, cosine_beta_schedule, extract,  # noqa: F401
                         linear_beta_schedule, sigmoid_beta_schedule)
from ._unet import UnetBase


class Unet(UnetBase):
    def __init__(self, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=1, self_condition=False,
                 resnet_block_groups=8, learned_variance=False, learned_sinusoidal_cond=False,
                 random_fourier_features=False, learned_sinusoidal_dim=16):
        super().__init__(dim, init_dim, out_dim, dim_mults, channels, self_condition, resnet_block_groups,
                         learned_variance, learned_sinusoidal_cond, random_fourier_features, learned_sinusoidal_dim)


class GaussianDiffusion(DiffusionCore):
    KIND = "uncond"

    def p_losses(self, x_start, t, noise=None):
        """src/hicdiff.py:711-747.  Train mode under autograd: loss with .backward() from the native training step; else the value."""
        if noise is None:
            noise = self.noise_source.randn(x_start.shape) if self.noise_source is not None else torch.randn_like(x_start)
        if self.self_condition:
            raise NotImplementedError("self-conditioning on the model's own x0 (src/hicdiff.py:723-727) is unused by HiCDiff; "
                                      "use hicdiff_condition for conditioning on the low-coverage tile")
        if self._native_training():
            return self._native_loss(x_start, None, t, noise)
        x = self.q_sample(x_start=x_start, t=t, noise=noise)
        out = self.model(x, t, None)
        return self._loss_value(out, self._target(x_start, t, noise), t)

    def forward(self, img, *args, **kwargs):
        b, c, h, w = img.shape
        assert h == self.image_size and w == self.image_size, f"height and width of image must be {self.image_size}"
        t = torch.randint(0, self.num_timesteps, (b,), device=img.device).long()
        return self.p_losses(self.normalize(img), t, *args, **kwargs)
