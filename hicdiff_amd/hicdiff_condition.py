"""Conditional HiCDiff (low-coverage tile concatenated on the channel axis): drop-in for
``src/hicdiff_condition.py`` (p_sample_loop :600-623, super_resolution :676-678, p_losses :715-746)."""
from __future__ import annotations

import torch

from ._diffusion import (DiffusionCore, HostReplayNoise, ModelPrediction, cosine_beta_schedule, extract,  # noqa: F401
                         linear_beta_schedule, sigmoid_beta_schedule)
from ._unet import UnetBase


class Unet(UnetBase):
    def __init__(self, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=1, self_condition=True,
                 resnet_block_groups=8, learned_variance=False, learned_sinusoidal_cond=False,
                 random_fourier_features=False, learned_sinusoidal_dim=16):
        super().__init__(dim, init_dim, out_dim, dim_mults, channels, self_condition, resnet_block_groups,
                         learned_variance, learned_sinusoidal_cond, random_fourier_features, learned_sinusoidal_dim)


class GaussianDiffusion(DiffusionCore):
    KIND = "cond"

    @torch.no_grad()
    def p_sample_loop(self, x_in, return_all_timesteps=False):
        """x_in is the low-coverage batch when the model is conditional, else a shape tuple.
        return_all_timesteps gives the list [x_in | x_T, x_{T-1}, ..., x_0] as upstream."""
        if self.self_condition:
            if not torch.is_tensor(x_in):
                raise TypeError("conditional sampling needs the low-coverage tiles; sample()/shape tuples are the unconditional form")
            cond = x_in.contiguous().float()
            img, imgs = self._ancestral(tuple(cond.shape), cond, return_all_timesteps, cond)
        else:
            img, imgs = self._ancestral(tuple(x_in), None, return_all_timesteps, None)
        ret = img if not return_all_timesteps else imgs
        return self.unnormalize(ret) if not return_all_timesteps else ret

    @torch.no_grad()
    def super_resolution(self, x_in, continous=False):
        return self.p_sample_loop(x_in, continous)

    def p_losses(self, x_in, t=None, noise=None):
        x_start, x_end = x_in
        b, c, h, w = x_end.shape
        assert h == self.image_size and w == self.image_size, f"height and width of image must be {self.image_size}"
        if t is None:   # upstream always redraws t here (src/hicdiff_condition.py:719)
            t = torch.randint(0, self.num_timesteps, (b,), device=x_end.device).long()
        if noise is None:
            noise = self.noise_source.randn(x_end.shape) if self.noise_source is not None else torch.randn_like(x_end)
        if self._native_training():
            return self._native_loss(x_end, x_start if self.self_condition else None, t, noise)
        x = self.q_sample(x_start=x_end, t=t, noise=noise)
        out = self.model(x, t, x_start if self.self_condition else None)
        return self._loss_value(out, self._target(x_end, t, noise), t)

    def forward(self, img, *args, **kwargs):
        return self.p_losses([self.normalize(img[0]), self.normalize(img[1])], *args, **kwargs)
