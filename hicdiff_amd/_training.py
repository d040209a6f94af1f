"""Host side of the native training step (include/hicdiff_hip.h "training step").

The reference trains with `loss = diffusion(x); loss.backward(); optimizer.step(); optimizer.zero_grad()` (train.py:131-134).
Here the same four lines run on the HIP engine: in train mode `GaussianDiffusion.forward` computes the loss AND every gradient in
one `hd_train_loss_backward`, and returns a loss tensor whose `.backward()` hands the gradients to the parameters' `.grad`;
`hicdiff_amd.optim.Adam.step()` is one `hd_adam_step` over the flat buffer.  The module's parameters become views of one flat
fp32 tensor (state_dict keys, shapes and values unchanged).  torch owns the memory and, for N > 1 GPUs, the all-reduce:
`StagedReducer` sums the gradients over the ranks one gradient stage (hd_train_stage_*) at a time on a side stream while the
kernels of the later stages still run.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os

import torch

from . import _lib as L


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p()


def _dist_world() -> int:
    d = torch.distributed
    return d.get_world_size() if d.is_available() and d.is_initialized() else 1


class StagedReducer:
    """Data-parallel gradient sum over the ranks, overlapped with the walk back through the network.

    The C trainer finishes its gradients back to front and records one event per gradient stage (include/hicdiff_hip.h "Gradient
    stages").  For every stage, in stage order: a side stream waits for the stage's event (on the device), gathers the stage's
    slots of the flat gradient buffer into one contiguous bucket, and starts one asynchronous all-reduce of the bucket (RCCL on
    the GPU; the collectives of successive buckets queue behind each other while the compute stream is still producing the next
    stage).  `finish()` scatters the sums back over the flat buffer and makes the current stream wait for all of it; the host
    never blocks.  One bucket per stage rather than one collective per slot: xGMI ring all-reduces are latency-bound below a few
    MB, and a stage of either network is 10-40 MB.  Every rank builds the same buckets in the same order (the stage map is a
    function of the architecture alone), which is what the collectives need.

    On CPU tensors (the gloo tests) there are no streams: the same packing, collectives and scatter run in program order."""

    def __init__(self, grads: torch.Tensor, slots, slot_stage, group=None):
        self.grads, self.group = grads, group
        nst = max(slot_stage) + 1
        self.runs = [[] for _ in range(nst)]                     # per stage: merged (offset, length) runs of the flat buffer
        for (name, off, shape), k in sorted(zip(slots, slot_stage), key=lambda e: e[0][1]):
            n = 1
            for d in shape:
                n *= d
            runs = self.runs[k]
            if runs and runs[-1][0] + runs[-1][1] == off:
                runs[-1] = (runs[-1][0], runs[-1][1] + n)
            else:
                runs.append((off, n))
        self.buckets = [torch.empty(sum(n for _, n in runs), dtype=grads.dtype, device=grads.device) for runs in self.runs]
        self.cuda = grads.device.type == "cuda"
        self.comm = torch.cuda.Stream(device=grads.device) if self.cuda else None
        self.works = []

    @property
    def pending(self) -> bool:
        return bool(self.works)

    def _copy(self, k: int, to_bucket: bool):
        pos, b = 0, self.buckets[k]
        for off, n in self.runs[k]:
            if to_bucket:
                b[pos:pos + n].copy_(self.grads[off:off + n])
            else:
                self.grads[off:off + n].copy_(b[pos:pos + n])
            pos += n

    def launch(self, wait_stage=None):
        """Queue pack + all-reduce of every stage.  wait_stage(k, stream_handle) makes the side stream wait for stage k's event."""
        if self.works:
            raise RuntimeError("StagedReducer.launch: the previous reduction was not finished")
        for k in range(len(self.runs)):
            if not self.runs[k]:
                continue
            with (torch.cuda.stream(self.comm) if self.cuda else contextlib.nullcontext()):
                if wait_stage is not None:
                    wait_stage(k, self.comm.cuda_stream if self.cuda else None)
                self._copy(k, True)
                self.works.append((k, torch.distributed.all_reduce(self.buckets[k], group=self.group, async_op=True)))

    def finish(self):
        """Sums back into the flat buffer; the current stream continues behind them."""
        if not self.works:
            return
        with (torch.cuda.stream(self.comm) if self.cuda else contextlib.nullcontext()):
            for k, w in self.works:
                w.wait()                                         # GPU: the side stream waits for the collective's stream; CPU: blocks
                self._copy(k, False)
        self.works = []
        if self.cuda:
            torch.cuda.current_stream(self.grads.device).wait_stream(self.comm)


class NativeTrainer:
    """Flat parameter / gradient storage of one eps-network plus the C trainer sized for (B, S)."""

    def __init__(self, model, B: int, S: int, precision: str = "bf16x3"):
        p0 = next(model.parameters())
        if p0.device.type != "cuda":
            raise RuntimeError("hicdiff_amd trains on MI355X (HIP) tensors only; there is no CPU fallback")
        self.lib, self.model, self.device, self.B, self.S = L.load(), model, p0.device, B, S
        self.h = C.c_void_p()
        arch = model._arch()
        with torch.cuda.device(self.device):
            rc = self.lib.hd_train_create(C.byref(self.h), self.device.index or 0, C.byref(arch), B, S)
        if rc != 0:
            msg = (self.lib.hd_train_last_error(None) or b"").decode()
            if rc == L.HD_EINVAL:
                raise NotImplementedError(msg or "native training is not available for this network")
            raise L.HdError(rc, msg)
        if precision not in ("bf16x3", "bf16"):
            raise ValueError("training precision is 'bf16x3' (default, fp32-equivalent products) or 'bf16'")
        self.precision = precision
        if precision == "bf16":
            self.lib.hd_train_set_precision(self.h, L.HD_TRAIN_PREC_BF16)
        total = C.c_longlong()
        n = self.lib.hd_train_param_count(self.h, C.byref(total))
        self.slots = []
        for i in range(n):
            name, off, shape, nd = C.c_char_p(), C.c_longlong(), (C.c_longlong * 4)(), C.c_int()
            self.lib.hd_train_param_slot(self.h, i, C.byref(name), C.byref(off), shape, C.byref(nd))
            self.slots.append((name.value.decode(), off.value, tuple(shape[k] for k in range(nd.value))))
        self.flat = torch.zeros(total.value, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros_like(self.flat)
        self.loss = torch.zeros((), dtype=torch.float32, device=self.device)
        named = dict(model.named_parameters())
        if set(named) != {s[0] for s in self.slots}:
            raise RuntimeError("parameter names of the module and of the trainer layout differ")
        self.params = []
        for name, off, shape in self.slots:                      # re-seat every parameter on the flat buffer
            p = named[name]
            if tuple(p.shape) != shape:
                raise RuntimeError(f"{name}: module shape {tuple(p.shape)} != trainer shape {shape}")
            view = self.flat[off:off + p.numel()].view(shape)
            view.copy_(p.detach())
            p.data = view
            p._hd_flat = (self, off)
            self.params.append(p)
        self.anchor = torch.zeros((), device=self.device, requires_grad=True)      # gives the returned loss a grad_fn
        stage = C.c_int()
        self.slot_stage = []
        for i in range(n):
            if self.lib.hd_train_slot_stage(self.h, i, C.byref(stage)) != 0:
                raise L.HdError(L.HD_EINVAL, "hd_train_slot_stage")
            self.slot_stage.append(stage.value)
        if max(self.slot_stage) + 1 != self.lib.hd_train_stage_count(self.h):
            raise RuntimeError("hd_train_stage_count disagrees with the slot -> stage map")
        self.reducer = None                # StagedReducer, made on the first step that runs under torch.distributed with > 1 rank
        self.reduced_serial = -1           # serial of the last loss whose gradients were (or are being) summed over the ranks

    def __del__(self):
        try:
            if getattr(self, "h", None) and self.h.value:
                self.lib.hd_train_destroy(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass

    def still_seated(self) -> bool:
        """False after e.g. load_state_dict(assign=True) or .to(): the parameters left the flat buffer."""
        base = self.flat.data_ptr()
        return all(p.data_ptr() == base + 4 * off for p, (_, off, _) in zip(self.params, self.slots))

    def grad_view(self, i):
        name, off, shape = self.slots[i]
        n = 1
        for s in shape:
            n *= s
        return self.grads[off:off + n].view(shape)

    def loss_backward(self, x_start, cond, t, noise, a_t, s_t, l2: bool, objective: str = "pred_noise", loss_weights=None):
        """loss_weights: per-sample weights of the loss (p2_loss_weight[t], src/hicdiff.py:746), or None for the plain mean."""
        B, _, S, _ = x_start.shape
        self._lw = None if loss_weights is None else loss_weights.detach().to(torch.float32).contiguous()      # kept alive until the next call
        rc = self.lib.hd_train_set_loss_weights(self.h, None if self._lw is None else C.c_void_p(self._lw.data_ptr()))
        if rc != 0:
            raise L.HdError(rc, (self.lib.hd_train_last_error(self.h) or b"").decode())
        if objective != getattr(self, "objective", "pred_noise"):
            rc = self.lib.hd_train_set_objective(self.h, {"pred_noise": 0, "pred_x0": 1, "pred_v": 2}[objective])
            if rc != 0:
                raise L.HdError(rc, (self.lib.hd_train_last_error(self.h) or b"").decode())
            self.objective = objective
        if (B, S) != (self.B, self.S):
            raise ValueError(f"trainer sized for batches of {self.B} tiles of {self.S}x{self.S}, got {B} of {S}x{S}")
        f = lambda v: None if v is None else v.detach().to(torch.float32).contiguous()
        x_start, cond, noise, a_t, s_t = f(x_start), f(cond), f(noise), f(a_t), f(s_t)
        if t.dtype.is_floating_point:                       # SR3: the continuous noise level
            t, kind = t.detach().to(torch.float32).reshape(-1).contiguous(), L.HD_T_FLOAT32
        else:
            t, kind = t.to(torch.int64).contiguous(), L.HD_T_INT64
        # Gradient accumulation (several diffusion(x) / backward() pairs before one optimizer.step(), no zero_grad between): the
        # kernels overwrite the flat gradient buffer, and after the first backward() every p.grad IS a view of that buffer.  Carry
        # the gradients accumulated so far in a copy; backward() adds them back (as autograd's AccumulateGrad would).
        self.reduce_finish()               # a reduction of the previous loss still in flight writes the buffer this step overwrites
        if getattr(self, "carry", None) is None and any(
                p.grad is not None and p.grad.data_ptr() == self.grad_view(i).data_ptr() for i, p in enumerate(self.params)):
            self.carry = self.grads.clone()
        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            rc = self.lib.hd_train_loss_backward(self.h, _ptr(self.flat), _ptr(self.grads), _ptr(x_start), _ptr(cond), _ptr(t), kind, _ptr(noise),
                                                 _ptr(a_t), _ptr(s_t), 1 if l2 else 0, _ptr(self.loss), st)
        if rc != 0:
            raise L.HdError(rc, (self.lib.hd_train_last_error(self.h) or b"").decode() + " / " + (self.lib.hd_last_error(None) or b"").decode())
        self.serial = getattr(self, "serial", 0) + 1
        if _dist_world() > 1 and os.environ.get("HICDIFF_DP_OVERLAP", "1") != "0":
            # every kernel of the step is queued; queue the per-stage sums behind their events now, so that stage k travels over xGMI
            # while stages k+1.. are still being computed.  (HICDIFF_DP_OVERLAP=0: one all-reduce of the whole buffer in Adam.step.)
            # REQUIREMENT: the collectives are issued HERE, in the forward call -- unlike DDP, which communicates in backward() -- so under
            # torchrun every rank must make the same sequence of train-mode, grad-enabled `diffusion(x)` calls (no rank-0-only loss probe,
            # equal batch counts per epoch: train.py trims ragged shards).  Evaluate under torch.no_grad() / .eval() for anything else.
            if self.reducer is None:
                self.reducer = StagedReducer(self.grads, self.slots, self.slot_stage)
            with torch.cuda.device(self.device):
                self.reducer.launch(self._wait_stage)
            self.reduced_serial = self.serial
        return _NativeLoss.apply(self.anchor, self, self.loss.clone(), self.serial)

    def _wait_stage(self, k, stream):
        rc = self.lib.hd_train_stage_wait(self.h, k, C.c_void_p(stream))
        if rc != 0:
            raise L.HdError(rc, (self.lib.hd_train_last_error(self.h) or b"").decode())

    def reduce_finish(self):
        if self.reducer is not None and self.reducer.pending:
            with torch.cuda.device(self.device):
                self.reducer.finish()

    def weights_changed(self):
        self.model.__dict__["_hd_weight_epoch"] = self.model.__dict__.get("_hd_weight_epoch", 0) + 1


class _NativeLoss(torch.autograd.Function):
    """The gradients already exist when the loss is returned; backward() only publishes them as `.grad`."""

    @staticmethod
    def forward(ctx, anchor, trainer, value, serial):
        ctx.trainer, ctx.serial = trainer, serial
        return value

    @staticmethod
    def backward(ctx, grad_out):
        tr = ctx.trainer
        if ctx.serial != tr.serial:
            raise RuntimeError("this loss is stale: the gradient buffer holds the gradients of a later diffusion(x) call "
                               "(call loss.backward() before the next forward in train mode)")
        tr.reduce_finish()                        # the sums over the ranks (if any) land before anything below touches the buffer
        one = bool((grad_out == 1).item())
        if not one:
            tr.grads.mul_(grad_out)
        carry, tr.carry = getattr(tr, "carry", None), None
        for i, p in enumerate(tr.params):
            g = tr.grad_view(i)
            if p.grad is None:
                p.grad = g                        # a view of the flat gradient buffer: Adam below reads it in place
            elif p.grad.data_ptr() != g.data_ptr():
                p.grad.add_(g)                    # a .grad tensor of the caller's own: accumulate into it, as autograd would
            elif carry is not None:
                g.add_(carry[tr.slots[i][1]:tr.slots[i][1] + g.numel()].view_as(g))    # the live view: add what it held before this forward
        return None, None, None, None


def trainer_for(model, B: int, S: int) -> NativeTrainer:
    """The network's trainer, re-created when the batch shape, the device or the requested arithmetic changed.  The arithmetic is
    `model.train_precision` ('bf16x3' | 'bf16') if set, else the environment's HICDIFF_TRAIN_PRECISION, else 'bf16x3'."""
    precision = getattr(model, "train_precision", None) or os.environ.get("HICDIFF_TRAIN_PRECISION", "bf16x3")
    tr = model.__dict__.get("_hd_trainer")
    if tr is None or (tr.B, tr.S) != (B, S) or tr.device != next(model.parameters()).device or not tr.still_seated() or tr.precision != precision:
        if tr is not None:
            tr.reduce_finish()          # a staged reduction of the old trainer's last loss may still be writing its gradient buffer
            for p in tr.params:
                p.grad = None
        tr = NativeTrainer(model, B, S, precision)
        model.__dict__["_hd_trainer"] = tr
    return tr
