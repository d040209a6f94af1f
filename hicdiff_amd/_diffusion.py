"""Host side of the DDPM sampler / loss around the HIP engine.

One implementation serves the reference's three near-identical ``GaussianDiffusion`` classes
(src/hicdiff.py:432-755 unconditional, src/hicdiff_condition.py:429-750 conditional,
src/hicdiff_sr3.py:488-796 SR3); ``hicdiff.py`` / ``hicdiff_condition.py`` / ``hicdiff_sr3.py`` in
this package bind the reference's defaults.  The per-step work (epsilon-network + posterior update
+ noise) is one C-ABI call, ``hd_ddpm_step``; this file only gathers the per-step scalars from the
schedule buffers and keeps the reference's method names, argument meaning and assertions.
"""
from __future__ import annotations

import contextlib
import math
import os
from collections import namedtuple
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import _lib as L

ModelPrediction = namedtuple("ModelPrediction", ["pred_noise", "pred_x_start"])


# ---- beta schedules, fp64 (src/hicdiff.py:396-430) ---------------------------------------------

def linear_beta_schedule(timesteps):
    scale = 1000 / timesteps
    return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)


def cosine_beta_schedule(timesteps, s=0.008):
    t = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64) / timesteps
    ac = torch.cos((t + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    return torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)


def sigmoid_beta_schedule(timesteps, start=-3, end=3, tau=1, clamp_min=1e-5):
    t = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64) / timesteps
    v_start = torch.tensor(start / tau).sigmoid()
    v_end = torch.tensor(end / tau).sigmoid()
    ac = (-((t * (end - start) + start) / tau).sigmoid() + v_end) / (v_end - v_start)
    ac = ac / ac[0]
    return torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)


_SCHEDULES = {"linear": linear_beta_schedule, "cosine": cosine_beta_schedule, "sigmoid": sigmoid_beta_schedule}


def extract(a, t, x_shape):
    """a.gather(-1, t) broadcast to x (src/hicdiff.py:391-394)."""
    out = a.gather(-1, t)
    return out.reshape(t.shape[0], *((1,) * (len(x_shape) - 1)))


# ---- noise sources ----------------------------------------------------------------------------

class HostReplayNoise:
    """Parity runs: draw every Gaussian tensor from a torch CPU generator in the reference's call
    order (``torch.randn(shape)`` then one ``randn_like`` per step, src/hicdiff.py:607,599) and copy
    it to the device, so a seeded reference run can be replayed bit-for-bit in its noise."""

    def __init__(self, seed: int, device):
        self.gen = torch.Generator(device="cpu")
        self.gen.manual_seed(int(seed))
        self.device = device

    def randn(self, shape):
        return torch.randn(tuple(shape), generator=self.gen, dtype=torch.float32).to(self.device)


class DiffusionCore(nn.Module):
    KIND = "uncond"          # 'uncond' | 'cond' | 'sr3'

    def __init__(self, model, *, image_size, timesteps=1000, sampling_timesteps=None, loss_type="l1",
                 objective="pred_noise", beta_schedule="sigmoid", schedule_fn_kwargs=dict(),
                 p2_loss_weight_gamma=0., p2_loss_weight_k=1, ddim_sampling_eta=0., auto_normalize=False):
        super().__init__()
        # same guards as src/hicdiff.py:450-451,461,470,486
        assert not (self.KIND == "uncond" and type(self).__name__ == "GaussianDiffusion" and model.channels != model.out_dim)
        assert not model.random_or_learned_sinusoidal_cond
        assert objective in {"pred_noise", "pred_x0", "pred_v"}, \
            "objective must be either pred_noise (predict noise) or pred_x0 (predict image start) or pred_v (predict v)"
        if beta_schedule not in _SCHEDULES:
            raise ValueError(f"unknown beta schedule {beta_schedule}")
        self.model = model
        self.channels = model.channels
        self.self_condition = model.self_condition
        self.image_size = image_size
        self.objective = objective
        self.loss_type = loss_type

        betas = _SCHEDULES[beta_schedule](timesteps, **schedule_fn_kwargs)
        self.__dict__["_beta_schedule_name"] = beta_schedule          # the precision schedule is measured per beta schedule (_early_band)
        alphas = 1. - betas
        ac = torch.cumprod(alphas, dim=0)
        ac_prev = F.pad(ac[:-1], (1, 0), value=1.)
        self.num_timesteps = int(betas.shape[0])
        self.sampling_timesteps = sampling_timesteps if sampling_timesteps is not None else timesteps
        assert self.sampling_timesteps <= timesteps
        self.is_ddim_sampling = self.sampling_timesteps < timesteps
        self.ddim_sampling_eta = ddim_sampling_eta
        if self.KIND == "sr3":
            # plain fp64 attribute, length T+1, first two entries 1.0 (src/hicdiff_sr3.py:535-536)
            self.sqrt_alphas_cumprod_prev = torch.sqrt(F.pad(ac_prev, (1, 0), value=1.))

        post_var = betas * (1. - ac_prev) / (1. - ac)
        table = {
            "betas": betas,
            "alphas_cumprod": ac,
            "alphas_cumprod_prev": ac_prev,
            "sqrt_alphas_cumprod": torch.sqrt(ac),
            "sqrt_one_minus_alphas_cumprod": torch.sqrt(1. - ac),
            "log_one_minus_alphas_cumprod": torch.log(1. - ac),
            "sqrt_recip_alphas_cumprod": torch.sqrt(1. / ac),
            "sqrt_recipm1_alphas_cumprod": torch.sqrt(1. / ac - 1),
            "posterior_variance": post_var,
            "posterior_log_variance_clipped": torch.log(post_var.clamp(min=1e-20)),
            "posterior_mean_coef1": betas * torch.sqrt(ac_prev) / (1. - ac),
            "posterior_mean_coef2": (1. - ac_prev) * torch.sqrt(alphas) / (1. - ac),
            "p2_loss_weight": (p2_loss_weight_k + ac / (1 - ac)) ** -p2_loss_weight_gamma,
        }
        for name, val in table.items():           # the 13 fp32 buffers of src/hicdiff.py:494-522
            self.register_buffer(name, val.to(torch.float32))
        self.__dict__["_host_cache"] = None       # host copies of the buffers, see _host

        self.normalize = (lambda img: img * 2 - 1) if auto_normalize else (lambda img: img)
        self.unnormalize = (lambda t: (t + 1) * 0.5) if auto_normalize else (lambda t: t)

        # sampling knobs that do not exist upstream
        # Precision schedule of the ancestral chain: in the first half of a long chain (t >= T / 2, T >= 1000) the
        # 3x3 convolutions run on two fp16 products per multiply instead of three bf16 ones, in its first quarter on one -- the chain damps what that costs
        # (tests/studies/error_budget_study.py; tests/test_gpu_timed_path.py::test_full_length_chain_drift_vs_oracle holds the bound).
        # False: split-bf16 x3 at every step.  HICDIFF_EARLY_F16=0 turns it off for a process.
        self.early_band_f16 = os.environ.get("HICDIFF_EARLY_F16", "1") != "0"
        # the band is t >= early_band_from * T; None: the network's own figure (UNet 0.5, hicedrn 0: profiles/r04_e_*, r04_p_*)
        self.early_band_from = None
        # below the band: two products on the 3x3 layers of the feature maps of at most (S/4)^2 pixels -- the study's "low" layer class, 42 % of the
        # matrix work, which the late half of the chain tolerates where the full-resolution layers do not (profiles/r04_m_*); HICDIFF_LATE_LOW_F16=0: off
        self.late_band_low_f16 = os.environ.get("HICDIFF_LATE_LOW_F16", "1") != "0"
        self.early_band_x1_from = 0.75  # inside the band, t >= early_band_x1_from * T takes ONE fp16 product, xh wh (> 1: never); measured: r04_j
        self.noise_source = None     # None: device Philox; or an object with .randn(shape) -> device tensor
        self.seed = 1234             # Philox key for device noise
        self.tile_offset = 0         # global index of this rank's first tile (sharded sampling)

    _HOST_KEYS = ("alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1",
                  "posterior_mean_coef2", "posterior_log_variance_clipped", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod")

    @property
    def _host(self):
        """Host copies of the schedule buffers the fused steps take their per-step scalars from (no device sync per step).
        The reference samples from whatever its registered buffers hold -- a checkpoint (`torch.save(diffusion.state_dict())`,
        train.py:186) carries all 13 -- so the copies follow the buffers: they are rebuilt whenever a buffer was written
        (load_state_dict copies in place and bumps `_version`) or replaced (.to(), assign=True)."""
        bufs = [getattr(self, k) for k in self._HOST_KEYS]
        sig = tuple((b.data_ptr(), b._version, b.device) for b in bufs)
        cache = self.__dict__.get("_host_cache")
        if cache is None or cache[0] != sig:
            host = {k: b.detach().to("cpu", torch.float32).clone() for k, b in zip(self._HOST_KEYS, bufs)}
            host["sigma"] = (0.5 * host["posterior_log_variance_clipped"]).exp()
            cache = (sig, host)
            self.__dict__["_host_cache"] = cache
        return cache[1]

    # -- small algebra, kept for API parity (device torch ops on gathered scalars) --------------
    def predict_start_from_noise(self, x_t, t, noise):
        return extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t - extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * noise

    def predict_noise_from_start(self, x_t, t, x0):
        return (extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t - x0) / extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape)

    def predict_v(self, x_start, t, noise):
        return extract(self.sqrt_alphas_cumprod, t, x_start.shape) * noise - extract(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * x_start

    def predict_start_from_v(self, x_t, t, v):
        return extract(self.sqrt_alphas_cumprod, t, x_t.shape) * x_t - extract(self.sqrt_one_minus_alphas_cumprod, t, x_t.shape) * v

    def q_posterior(self, x_start, x_t, t):
        mean = extract(self.posterior_mean_coef1, t, x_t.shape) * x_start + extract(self.posterior_mean_coef2, t, x_t.shape) * x_t
        return mean, extract(self.posterior_variance, t, x_t.shape), extract(self.posterior_log_variance_clipped, t, x_t.shape)

    def _time_arg(self, t_idx: int, batch: int, device):
        if self.KIND == "sr3":
            level = float(np.float32(self.sqrt_alphas_cumprod_prev[t_idx + 1].item()))
            return torch.full((batch, 1), level, device=device, dtype=torch.float32)
        return torch.full((batch,), t_idx, device=device, dtype=torch.long)

    def model_predictions(self, x, t, x_self_cond=None, clip_x_start=False, t_real=None):
        """src/hicdiff.py:562-582 (SR3: t is the noise level, t_real the integer step)."""
        out = self.model(x, t, x_self_cond)
        tt = t if t_real is None else torch.full((x.shape[0],), int(t_real), device=x.device, dtype=torch.long)
        clip = (lambda v: v.clamp(-1., 1.)) if clip_x_start else (lambda v: v)
        if self.objective == "pred_noise":
            pred_noise = out
            x_start = clip(self.predict_start_from_noise(x, tt, out))
        elif self.objective == "pred_x0":
            x_start = clip(out)
            pred_noise = self.predict_noise_from_start(x, tt, x_start)
        else:
            x_start = clip(self.predict_start_from_v(x, tt, out))
            pred_noise = self.predict_noise_from_start(x, tt, x_start)
        return ModelPrediction(pred_noise, x_start)

    def p_mean_variance(self, x, t, x_self_cond=None, clip_denoised=True):
        if self.KIND == "sr3":
            t_idx = int(t)
            preds = self.model_predictions(x, self._time_arg(t_idx, x.shape[0], x.device), x_self_cond, t_real=t_idx)
            t = torch.full((x.shape[0],), t_idx, device=x.device, dtype=torch.long)
        else:
            preds = self.model_predictions(x, t, x_self_cond)
        x_start = preds.pred_x_start
        if clip_denoised:
            x_start = x_start.clamp(-1., 1.)
        mean, var, logvar = self.q_posterior(x_start=x_start, x_t=x, t=t)
        return mean, var, logvar, x_start

    # -- fused reverse step ----------------------------------------------------------------------
    def _x0_coefs(self, t_idx: int):
        """(f1, f2) with x0 = f1 x - f2 out, the form the fused update kernel evaluates (hd_ddpm_coef's first two fields): what the network
        predicts is the noise (predict_start_from_noise, src/hicdiff.py:529-533), x0 itself, or v (predict_start_from_v, :548-552)."""
        h = self._host
        if self.objective == "pred_noise":
            return float(h["sqrt_recip_alphas_cumprod"][t_idx]), float(h["sqrt_recipm1_alphas_cumprod"][t_idx])
        if self.objective == "pred_x0":
            return 0.0, -1.0
        return float(h["sqrt_alphas_cumprod"][t_idx]), float(h["sqrt_one_minus_alphas_cumprod"][t_idx])

    def _coef(self, t_idx: int) -> L.HdDdpmCoef:
        h = self._host
        c = L.HdDdpmCoef()
        c.sqrt_recip_alphas_cumprod, c.sqrt_recipm1_alphas_cumprod = self._x0_coefs(t_idx)
        c.posterior_mean_coef1 = float(h["posterior_mean_coef1"][t_idx])
        c.posterior_mean_coef2 = float(h["posterior_mean_coef2"][t_idx])
        c.sigma = float(h["sigma"][t_idx]) if t_idx > 0 else 0.0
        if self.KIND == "sr3":
            c.time_value = float(np.float32(self.sqrt_alphas_cumprod_prev[t_idx + 1].item()))
        else:
            c.time_value = float(t_idx)
        c.arith = L.HD_ARITH_DEFAULT
        if self.late_band_low_f16 and self._early_band(self.num_timesteps - 1):
            c.arith = L.HD_ARITH_F16W2_LOW
        if self._early_band(t_idx):
            c.arith = L.HD_ARITH_F16W1 if t_idx >= int(self.early_band_x1_from * self.num_timesteps) else L.HD_ARITH_F16W2
        return c

    def _early_band(self, t_idx: int) -> bool:
        """Does step t take the two-product arithmetic?  Chains of at least 1000 steps only -- the lengths measured (1000; SR3's 2000); a 50-step
        chain amplifies a per-step error 4-7 x more (each of its steps is twenty steps' worth of posterior_mean_coef1) and a 500-step one would
        sit in between with no measurement behind it --, their first half only, networks that declare EARLY_BAND_OK (what the CPU
        study and the GPU drift runs covered) -- and the LINEAR beta schedule only (EARLY_BAND_SCHEDULES): what a chain forgives depends on how
        long it stays at low noise.  The same bands on the reference's default sigmoid schedule (alphas_cumprod 0.50 at T/2 against the linear
        schedule's 0.078) and on the cosine one ended 1.0-2.4e-3 from the x3 chain, and bands moved to where those chains are as noisy as
        the linear one (t >= 835 / 989 of 1000) with two products on the low-resolution layers below still 0.3-1.9e-3: half of a sigmoid
        chain runs at alphas_cumprod > 0.5, where nothing damps an error, against a sixth of a linear one (profiles/r04_s_*).  Other
        schedules keep split-bf16 x3 at every step (1.0-1.9e-4 from the oracle on the same chains).  And the pred_noise objective only -- the
        reference's, and the one measured: with pred_x0 / pred_v the network's error reaches x0 with gain 1 instead of
        sqrt_recipm1_alphas_cumprod[t], which is 0.01 at the end of a chain, exactly where the late tier would put it."""
        T = self.num_timesteps
        return (bool(self.early_band_f16) and T >= 1000 and t_idx >= int(self._band_from() * T) and bool(getattr(self.model, "EARLY_BAND_OK", False))
                and self.__dict__.get("_beta_schedule_name") in self.EARLY_BAND_SCHEDULES and self.objective == "pred_noise")

    EARLY_BAND_SCHEDULES = ("linear",)     # beta schedules the precision schedule has been measured on (train.py's; BASELINE's bench configuration)

    def _band_from(self) -> float:
        return float(getattr(self.model, "EARLY_BAND_FROM", 0.5) if self.early_band_from is None else self.early_band_from)

    def _step_inplace(self, img, t_idx: int, cond, x0_out=None, eng=None):
        """img <- p_sample(img, t): one hd_ddpm_step call (eps-net + clamp + posterior + noise).  The three objectives differ only in the two
        coefficients that turn the network's output into x0 (_x0_coefs).  `eng`: the engine a loop looked up once -- `model.engine()` compares
        every parameter's (address, version) with the packed copy, 0.4 ms of host time that belongs to the chain, not to each step."""
        if eng is None:
            eng = self.model.engine(img.device)
        noise = None
        if t_idx > 0 and self.noise_source is not None:
            noise = self.noise_source.randn(img.shape).contiguous()
        eng.ddpm_step(img, cond, noise, self._coef(t_idx), x0_out, seed=self.seed, tile_offset=self.tile_offset, step=t_idx)

    def _initial_noise(self, shape, device):
        if self.noise_source is not None:
            return self.noise_source.randn(shape).contiguous()
        return self.model.engine(device).randn(shape[0], shape[2], self.seed, self.tile_offset, self.num_timesteps)

    @torch.no_grad()
    def p_sample(self, x, t, x_self_cond=None):
        """src/hicdiff.py:594-601: returns (x_{t-1}, clamped x0 estimate); x is left untouched."""
        t_idx = int(t)
        img = x.contiguous().float().clone()
        x0 = torch.empty_like(img)
        cond = x_self_cond.contiguous().float() if (self.self_condition and x_self_cond is not None) else None
        self._step_inplace(img, t_idx, cond, x0)
        return img, x0

    def _device(self):
        return self.betas.device

    def _bracket(self, eng, shape, free_running: bool):
        """A chain bracket (Engine.chain) when nothing on the caller's stream has to see the state between steps: device noise, no
        per-step copies.  Otherwise every step hands over to the caller's stream as before."""
        return eng.chain(shape[0], shape[-1]) if free_running else contextlib.nullcontext()

    @torch.no_grad()
    def _ancestral(self, shape, cond, return_all_timesteps, first):
        device = self._device()
        eng = self.model.engine(device)
        img = self._initial_noise(shape, device)
        imgs = [first if first is not None else img.clone()] if return_all_timesteps else None
        with self._bracket(eng, shape, self.noise_source is None and not return_all_timesteps):
            for t in reversed(range(self.num_timesteps)):
                self._step_inplace(img, t, cond, eng=eng)
                if return_all_timesteps:       # upstream keeps all T+1 tensors alive (src/hicdiff.py:615); here only on request
                    imgs.append(img.clone())
        return img, imgs

    @torch.no_grad()
    def p_sample_loop(self, shape, return_all_timesteps=False):
        """Unconditional form, src/hicdiff.py:603-620."""
        img, imgs = self._ancestral(tuple(shape), None, return_all_timesteps, None)
        ret = img if not return_all_timesteps else torch.stack(imgs, dim=1)
        return self.unnormalize(ret)

    @torch.no_grad()
    def ddim_sample(self, shape, return_all_timesteps=False):
        """src/hicdiff.py:622-664 on the fused step of the HIP engine."""
        eng = self.model.engine(self._device())
        saved = eng.precision
        # Strided DDIM divides eps by sqrt(alpha_bar) at a few, far-apart steps with no noise to wash the
        # difference out: a 2e-5 eps error can reach 1e-2 in x_0.  It runs 20-50x fewer network calls than
        # the ancestral chain, so it takes the exact-fp32 convolutions and keeps the 1e-3 parity bound.
        eng.set_precision(L.HD_PRECISION_F32)
        try:
            return self._ddim_sample(shape, return_all_timesteps)
        finally:
            eng.set_precision(saved)

    def _ddim_coef(self, time: int, time_next: int) -> L.HdDdpmCoef:
        """Coefficients that make the fused step a DDIM step: x <- sqrt(a_next) x0 + sqrt(1 - a_next - sigma^2) eps + sigma z
        (src/hicdiff.py:636-658)."""
        h, eta = self._host, self.ddim_sampling_eta
        ac = h["alphas_cumprod"]
        c = L.HdDdpmCoef()
        c.sqrt_recip_alphas_cumprod, c.sqrt_recipm1_alphas_cumprod = self._x0_coefs(time)
        c.time_value = float(time)
        c.posterior_mean_coef2 = 0.0
        if time_next < 0:                      # last step: x_0 itself (src/hicdiff.py:642-645)
            c.posterior_mean_coef1, c.eps_coef, c.sigma = 1.0, 0.0, 0.0
            return c
        a, an = ac[time], ac[time_next]
        sigma = eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
        c.posterior_mean_coef1, c.eps_coef, c.sigma = float(an.sqrt()), float((1 - an - sigma ** 2).sqrt()), float(sigma)
        if self.objective != "pred_noise":
            # the noise is derived from the CLIPPED x0 (model_predictions, src/hicdiff.py:571-580): eps = (R x - x0) / Rm1, so
            # sqrt(a_next) x0 + k eps = (sqrt(a_next) - k / Rm1) x0 + (k R / Rm1) x -- the kernel's x0 and x terms, no output term
            R, Rm1, k = float(h["sqrt_recip_alphas_cumprod"][time]), float(h["sqrt_recipm1_alphas_cumprod"][time]), c.eps_coef
            c.posterior_mean_coef1, c.posterior_mean_coef2, c.eps_coef = c.posterior_mean_coef1 - k / Rm1, k * R / Rm1, 0.0
        return c

    def _ddim_sample(self, shape, return_all_timesteps=False):
        """Each DDIM step is the same fused call as the ancestral step (hd_ddpm_step: eps-net + clamp + update + noise, hipGraph
        replay with device noise) with other coefficients (_ddim_coef)."""
        shape = tuple(shape)
        device, T, S = self._device(), self.num_timesteps, self.sampling_timesteps
        times = list(reversed(torch.linspace(-1, T - 1, steps=S + 1).int().tolist()))
        eng = self.model.engine(device)
        img = self._initial_noise(shape, device)
        imgs = [img.clone()] if return_all_timesteps else None
        with self._bracket(eng, shape, self.noise_source is None and not return_all_timesteps):
            for time, time_next in zip(times[:-1], times[1:]):
                noise = None
                if time_next >= 0 and self.noise_source is not None:      # the reference draws randn_like(img) at every such step, eta = 0 included
                    noise = self.noise_source.randn(shape).contiguous()
                eng.ddpm_step(img, None, noise, self._ddim_coef(time, time_next), None, seed=self.seed, tile_offset=self.tile_offset, step=time)
                if return_all_timesteps:
                    imgs.append(img.clone())
        ret = img if not return_all_timesteps else torch.stack(imgs, dim=1)
        return self.unnormalize(ret)

    @torch.no_grad()
    def sample(self, x, return_all_timesteps=False):
        """src/hicdiff.py:666-671: only ``x.shape[0]`` is used."""
        shape = (x.shape[0], self.channels, self.image_size, self.image_size)
        fn = self.p_sample_loop if not self.is_ddim_sampling else self.ddim_sample
        return fn(shape, return_all_timesteps=return_all_timesteps)

    @torch.no_grad()
    def interpolate(self, x1, x2, t=None, lam=0.5):
        """src/hicdiff.py:673-691."""
        b = x1.shape[0]
        t = self.num_timesteps - 1 if t is None else t
        assert x1.shape == x2.shape
        tb = torch.full((b,), t, device=x1.device, dtype=torch.long)
        img = ((1 - lam) * self.q_sample(x1, tb) + lam * self.q_sample(x2, tb)).contiguous()
        eng = self.model.engine(img.device)
        with self._bracket(eng, img.shape, self.noise_source is None):
            for i in reversed(range(0, t)):
                self._step_inplace(img, i, None, eng=eng)
        return img

    # -- forward process and loss ----------------------------------------------------------------
    def q_sample(self, x_start, t, noise=None):
        """src/hicdiff.py:694-700."""
        eng = self.model.engine(x_start.device)
        x_start = x_start.contiguous().float()
        if noise is None:
            noise = self.noise_source.randn(x_start.shape) if self.noise_source is not None else torch.randn_like(x_start)
        a = self.sqrt_alphas_cumprod.gather(-1, t).contiguous()
        s = self.sqrt_one_minus_alphas_cumprod.gather(-1, t).contiguous()
        return eng.q_sample(x_start, noise.contiguous().float(), a, s)

    @property
    def loss_fn(self):
        if self.loss_type == "l1":
            return F.l1_loss
        if self.loss_type == "l2":
            return F.mse_loss
        raise ValueError(f"invalid loss type {self.loss_type}")

    def _loss_value(self, model_out, target, t):
        if self.loss_type not in ("l1", "l2"):
            raise ValueError(f"invalid loss type {self.loss_type}")
        eng = self.model.engine(model_out.device)
        per = eng.loss_per_sample(model_out.contiguous(), target.contiguous().float(), self.loss_type == "l2")
        if t is not None:
            per = per * self.p2_loss_weight.gather(-1, t)
        return per.mean()

    def _native_training(self) -> bool:
        """Train mode under autograd -> the native training step (hicdiff_amd/_training.py); eval / no_grad -> loss value only."""
        return self.training and torch.is_grad_enabled() and bool(getattr(self.model, "_native_train", False))

    def _native_loss(self, x_start, cond, t, noise, level=None):
        """loss tensor with `.backward()` (train.py:131-132): forward, loss and every gradient in one engine call."""
        from ._training import trainer_for
        if self.loss_type not in ("l1", "l2"):
            raise ValueError(f"invalid loss type {self.loss_type}")
        if level is not None:                                            # SR3: x_t = level x0 + sqrt(1 - level^2) eps, plain mean
            tr = trainer_for(self.model, x_start.shape[0], x_start.shape[-1])
            level = level.reshape(-1).to(x_start.device, torch.float32)
            return tr.loss_backward(x_start, cond, level, noise, level, (1 - level ** 2).sqrt(), self.loss_type == "l2")
        if self.__dict__.get("_p2_is_one") is None:                      # checked once: it costs a device -> host read
            self.__dict__["_p2_is_one"] = float(self.p2_loss_weight.min()) == 1.0 and float(self.p2_loss_weight.max()) == 1.0
        tr = trainer_for(self.model, x_start.shape[0], x_start.shape[-1])
        a_t = self.sqrt_alphas_cumprod.gather(-1, t)
        s_t = self.sqrt_one_minus_alphas_cumprod.gather(-1, t)
        lw = None if self.__dict__["_p2_is_one"] else self.p2_loss_weight.gather(-1, t)      # p2_loss_weight_gamma != 0 (src/hicdiff.py:522,746)
        return tr.loss_backward(x_start, cond, t, noise, a_t, s_t, self.loss_type == "l2", self.objective, lw)      # target: _target's (src/hicdiff.py:733-741)

    def _target(self, x_start, t, noise):
        if self.objective == "pred_noise":
            return noise
        if self.objective == "pred_x0":
            return x_start
        if self.objective == "pred_v":
            return self.predict_v(x_start, t, noise)
        raise ValueError(f"unknown objective {self.objective}")
