"""Host-side handle on one hd_ctx: owns the device context of an epsilon-network module, keeps its
packed weights in step with the module's parameters, and exposes the hot-path calls on torch
tensors (torch is only the owner of device memory and streams here)."""
from __future__ import annotations

import contextlib
import ctypes as C
import math
from typing import Optional

import torch
from torch import nn

from . import _lib as L


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("hicdiff_amd runs on MI355X (HIP) tensors only; there is no CPU fallback for the hot path")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class Engine:
    def __init__(self, arch: L.HdArchDesc, device: torch.device):
        self.lib = L.load()
        self.device = device
        self.arch = arch
        self.ctx = C.c_void_p()
        idx = device.index if device.index is not None else torch.cuda.current_device()
        rc = self.lib.hd_create(C.byref(self.ctx), idx, C.byref(arch))
        if rc != 0:
            raise L.HdError(rc, (self.lib.hd_last_error(None) or b"").decode())
        self._sig = None
        self._reserved = (0, 0)
        import os
        self.precision = L.HD_PRECISION_F32 if os.environ.get("HICDIFF_PRECISION", "") == "f32" else L.HD_PRECISION_BF16X3

    def __del__(self):
        try:
            if getattr(self, "ctx", None) and self.ctx.value:
                self.lib.hd_destroy(self.ctx)
                self.ctx = C.c_void_p()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise L.HdError(rc, (self.lib.hd_last_error(self.ctx) or b"").decode())

    # -- weights ------------------------------------------------------------------------------
    def sync_weights(self, module: nn.Module):
        """Re-pack when any parameter was updated in place or replaced (optimizer step,
        load_state_dict): the reference recomputes weight standardisation every forward
        (src/hicdiff.py:89-97); here it happens once per weight version."""
        params = list(module.named_parameters())
        sig = tuple((p.data_ptr(), p._version) for _, p in params) + (module.__dict__.get("_hd_weight_epoch", 0),)
        if sig == self._sig:
            return
        keep, arr = [], (L.HdNamedTensor * len(params))()
        for i, (name, p) in enumerate(params):
            _require_device(p)
            d = _f32c(p.detach())
            keep.append(d)
            arr[i].name = name.encode()
            arr[i].data = d.data_ptr()
            arr[i].ndim = d.dim()
            for j, s in enumerate(d.shape):
                arr[i].shape[j] = s
        with torch.cuda.device(self.device):
            self._check(self.lib.hd_load_weights(self.ctx, arr, len(params), _stream()))
            torch.cuda.current_stream().synchronize()   # sources may be temporaries (keep) -- load time only
        self._sig = sig

    def reserve(self, B: int, S: int):
        if B <= self._reserved[0] and S <= self._reserved[1]:
            return
        B2, S2 = max(B, self._reserved[0]), max(S, self._reserved[1])
        self._check(self.lib.hd_reserve(self.ctx, B2, S2))
        self._reserved = (B2, S2)

    def set_precision(self, mode: int):
        """L.HD_PRECISION_F32 (exact fp32 MFMA), L.HD_PRECISION_BF16X3 (default, split-bf16 x3) or L.HD_PRECISION_F16W2 (BF16X3 with two
        fp16 products in the 3x3 convolutions: the early band's arithmetic as a context-wide mode, for tests and measurements)."""
        self._check(self.lib.hd_set_precision(self.ctx, int(mode)))
        self.precision = int(mode)

    def set_chains(self, n: int):
        """2: always cut a replayed step into two half-batch chains on two streams, 1: never, 0: the default (by the amount of work)."""
        self._check(self.lib.hd_set_chains(self.ctx, int(n)))

    def chains_for(self, B: int, S: int) -> int:
        return int(self.lib.hd_chains_for(self.ctx, B, S))

    @contextlib.contextmanager
    def chain(self, B: int, S: int):
        """The loop of a sampler (src/hicdiff.py:603-620) as a bracket: inside it the replayed steps meet the caller's stream only at the
        first step and at the end, so the two half-batch chains of a large batch advance independently (hd_chain_begin / hd_chain_end).
        The caller must leave the step tensors alone on its own stream until the bracket closes."""
        self.reserve(B, S)
        with torch.cuda.device(self.device):
            self._check(self.lib.hd_chain_begin(self.ctx, _stream()))
        try:
            yield self
        finally:
            with torch.cuda.device(self.device):
                self._check(self.lib.hd_chain_end(self.ctx, _stream()))

    def workspace_bytes(self, B: int, S: int) -> int:
        out = C.c_size_t()
        self._check(self.lib.hd_workspace_bytes(self.ctx, B, S, C.byref(out)))
        return out.value

    # -- hot path ------------------------------------------------------------------------------
    def eps(self, x, t, cond=None):
        _require_device(x, t, cond)
        B, ch, S, S2 = x.shape
        if ch != 1 or S != S2:
            raise ValueError("expected tiles of shape (B, 1, S, S)")
        x = _f32c(x)
        cond = None if cond is None else _f32c(cond)
        if t.dtype in (torch.int64, torch.int32, torch.int16, torch.uint8):
            t, kind = t.to(torch.int64).contiguous(), L.HD_T_INT64
        else:
            t, kind = _f32c(t).reshape(-1), L.HD_T_FLOAT32
        if t.numel() != B:
            raise ValueError("time must have one entry per tile")
        self.reserve(B, S)
        out = torch.empty_like(x)
        with torch.cuda.device(self.device):
            self._check(self.lib.hd_eps_forward(self.ctx, _ptr(x), _ptr(t), kind, _ptr(cond), _ptr(out), B, S, _stream()))
        return out

    def ddpm_step(self, x, cond, noise, coef: L.HdDdpmCoef, x0_out=None, seed=0, tile_offset=0, step=0):
        B, _, S, _ = x.shape
        self.reserve(B, S)
        with torch.cuda.device(self.device):
            self._check(self.lib.hd_ddpm_step(self.ctx, _ptr(x), _ptr(cond), _ptr(noise), C.byref(coef), _ptr(x0_out), B, S,
                                              seed, tile_offset, step, _stream()))

    def ddrm_step(self, x, y, z, coef: L.HdDdrmCoef, x0_out=None, seed=0, tile_offset=0, step=0):
        B, _, S, _ = x.shape
        self.reserve(B, S)
        with torch.cuda.device(self.device):
            self._check(self.lib.hd_ddrm_step(self.ctx, _ptr(x), _ptr(y), _ptr(z), C.byref(coef), _ptr(x0_out), B, S,
                                              seed, tile_offset, step, _stream()))

    def q_sample(self, x0, noise, a, s):
        B, _, S, _ = x0.shape
        out = torch.empty_like(x0)
        with torch.cuda.device(self.device):
            self._check(self.lib.hd_q_sample(self.ctx, _ptr(x0), _ptr(noise), _ptr(a), _ptr(s), _ptr(out), B, S, _stream()))
        return out

    def loss_per_sample(self, pred, target, l2: bool):
        B, _, S, _ = pred.shape
        out = torch.empty(B, device=pred.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._check(self.lib.hd_loss_per_sample(self.ctx, _ptr(pred), _ptr(target), int(l2), _ptr(out), B, S, _stream()))
        return out

    def randn(self, B, S, seed, tile_offset, step):
        out = torch.empty((B, 1, S, S), device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._check(self.lib.hd_randn(self.ctx, _ptr(out), B, S, seed, tile_offset, step, _stream()))
        return out


# ------------------------------------------------------------------------------------------------
# Parameter tree: reproduces the reference's state-dict key paths ("downs.0.0.block1.proj.weight",
# "mid_attn.fn.fn.to_qkv.weight", ...) from a flat spec list, so checkpoints written by the
# reference's torch.save(diffusion.state_dict()) (train.py:186) load with strict=True.

class _Node(nn.Module):
    """Pure container; the compute lives in the HIP engine, not in nn.Module.forward."""


def _init_tensor(kind: str, shape, fan_in: int) -> torch.Tensor:
    t = torch.empty(shape, dtype=torch.float32)
    if kind == "ones":
        return t.fill_(1.0)
    if kind == "zeros":
        return t.zero_()
    bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0     # torch's default Conv2d/Linear init range
    return t.uniform_(-bound, bound)


def build_param_tree(root: nn.Module, specs) -> None:
    """specs: iterable of (dotted_name, shape, kind, fan_in) in reference registration order."""
    for name, shape, kind, fan_in in specs:
        parts = name.split(".")
        mod = root
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, _Node())
            mod = mod._modules[p]
        mod.register_parameter(parts[-1], nn.Parameter(_init_tensor(kind, shape, fan_in)))


class EpsNetBase(nn.Module):
    """Common host side of both epsilon-networks: engine lifetime + the reference call signature
    ``model(x, time, x_self_cond=None)`` (src/hicdiff.py:345, src/model/hicedrn_Diff.py:267)."""

    def _arch(self) -> L.HdArchDesc:  # pragma: no cover - abstract
        raise NotImplementedError

    def engine(self, device=None) -> Engine:
        _require_device(next(self.parameters()))
        device = device or next(self.parameters()).device
        if device.type != "cuda":
            raise RuntimeError("hicdiff_amd runs on MI355X (HIP) tensors only; there is no CPU fallback for the hot path")
        eng = self.__dict__.get("_eng")
        if eng is None or eng.device != device:
            eng = Engine(self._arch(), device)
            self.__dict__["_eng"] = eng
        eng.sync_weights(self)
        return eng

    def forward(self, x, time, x_self_cond=None):
        if self.self_condition and x_self_cond is None:
            raise ValueError("self_condition=True: x_self_cond (the low-coverage tile) is required")
        if not self.self_condition:
            x_self_cond = None
        # Inference engine: the returned eps carries no autograd graph (training backward is the
        # next scope row).
        with torch.no_grad():
            return self.engine(x.device).eps(x, time, x_self_cond)
