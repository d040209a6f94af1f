"""SR3-style HiCDiff (continuous noise-level conditioning): drop-in for ``src/hicdiff_sr3.py``
(p_mean_variance :634-644, q_sample :735-739, p_losses :750-792)."""
from __future__ import annotations

import numpy as np
import torch

from ._diffusion import (DiffusionCore, HostReplayNoise, ModelPrediction, cosine_beta_schedule, extract,  # noqa: F401
                         linear_beta_schedule, sigmoid_beta_schedule)
from ._unet import UnetBase
from .hicdiff_condition import GaussianDiffusion as _CondDiffusion


class Unet(UnetBase):
    def __init__(self, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=1, self_condition=True,
                 resnet_block_groups=8, learned_variance=False, learned_sinusoidal_cond=False,
                 random_fourier_features=False, noise_level_emb=False, learned_sinusoidal_dim=16):
        super().__init__(dim, init_dim, out_dim, dim_mults, channels, self_condition, resnet_block_groups,
                         learned_variance, learned_sinusoidal_cond, random_fourier_features, learned_sinusoidal_dim,
                         noise_level_emb=noise_level_emb)


class GaussianDiffusion(_CondDiffusion):
    KIND = "sr3"

    def __init__(self, model, *, image_size, timesteps=1000, sampling_timesteps=None, loss_type="l2", objective="pred_noise",
                 beta_schedule="linear", **kw):
        super().__init__(model, image_size=image_size, timesteps=timesteps, sampling_timesteps=sampling_timesteps,
                         loss_type=loss_type, objective=objective, beta_schedule=beta_schedule, **kw)

    def q_sample(self, x_start, continuous_sqrt_alpha_cumprod, noise=None):
        """x_t = level * x0 + sqrt(1 - level^2) * eps with a per-sample continuous level (B,1,1,1)."""
        eng = self.model.engine(x_start.device)
        x_start = x_start.contiguous().float()
        if noise is None:
            noise = self.noise_source.randn(x_start.shape) if self.noise_source is not None else torch.randn_like(x_start)
        a = continuous_sqrt_alpha_cumprod.reshape(-1).to(x_start.device, torch.float32).contiguous()
        s = (1 - a ** 2).sqrt().contiguous()
        return eng.q_sample(x_start, noise.contiguous().float(), a, s)

    def draw_level(self, batch, rng=np.random):
        """t ~ U{1..T}, level ~ U[sqrt_ac_prev[t-1], sqrt_ac_prev[t]] from numpy's generator, as upstream."""
        t = rng.randint(1, self.num_timesteps + 1)
        return torch.FloatTensor(rng.uniform(self.sqrt_alphas_cumprod_prev[t - 1], self.sqrt_alphas_cumprod_prev[t], size=batch))

    def p_losses(self, x_in, t=None, noise=None, level=None):
        x_start, x_end = x_in
        b, c, h, w = x_end.shape
        assert h == self.image_size and w == self.image_size, f"height and width of image must be {self.image_size}"
        if level is None:
            level = self.draw_level(b)
        level = level.to(x_end.device).view(b, -1)
        if noise is None:
            noise = self.noise_source.randn(x_end.shape) if self.noise_source is not None else torch.randn_like(x_end)
        if self._native_training():
            return self._native_loss(x_end, x_start if self.self_condition else None, None, noise, level=level)
        x = self.q_sample(x_start=x_end, continuous_sqrt_alpha_cumprod=level.view(-1, 1, 1, 1), noise=noise)
        out = self.model(x, level, x_start if self.self_condition else None)
        return self._loss_value(out, self._target(x_end, None, noise), None)   # plain mean, no p2 weight (:786-791)
