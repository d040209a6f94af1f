"""`optim.Adam(diffusion.parameters(), lr=2e-5)` of train.py:111 on the HIP engine: one `hd_adam_step` over the flat
parameter buffer of the native trainer (hicdiff_amd/_training.py).  With torch.distributed initialised the flat gradient is
summed over ranks first -- stage by stage behind the backward pass (hicdiff_amd/_training.py StagedReducer), or, with
HICDIFF_DP_OVERLAP=0, by one RCCL all-reduce of the whole buffer here -- and the mean is taken inside the Adam kernel."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


class Adam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("the reference trains with plain Adam (train.py:111)")
        self.params = [p for p in params]
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.state = {}            # eps-network module -> (m, v, step); survives a re-sized trainer (ragged last batch)
        self.param_groups = [{"params": self.params, "lr": self.lr, "betas": self.betas, "eps": self.eps}]

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        trainers = []
        for p in self.params:
            tr = getattr(p, "_hd_flat", (None,))[0]
            if tr is None:
                if p.grad is not None:
                    raise RuntimeError("hicdiff_amd.optim.Adam steps parameters of a natively trained network only "
                                       "(call diffusion(x) in train mode first)")
                continue
            if tr not in trainers:
                trainers.append(tr)
        lib = L.load()
        for tr in trainers:
            if any(p.grad is None for p in tr.params):
                continue                                           # nothing was back-propagated since zero_grad
            for i, p in enumerate(tr.params):                      # gradients someone replaced: copy them into the flat buffer
                g = tr.grad_view(i)
                if p.grad.data_ptr() != g.data_ptr():
                    g.copy_(p.grad)
            m, v, k = self.state.get(tr.model, (None, None, 0))
            if m is None or m.numel() != tr.flat.numel() or m.device != tr.flat.device:
                m, v = torch.zeros_like(tr.flat), torch.zeros_like(tr.flat)
            k += 1
            scale = 1.0
            if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
                tr.reduce_finish()
                if tr.reduced_serial != tr.serial:                 # not summed stage by stage while the backward pass ran
                    torch.distributed.all_reduce(tr.grads)
                scale = 1.0 / torch.distributed.get_world_size()
            lr = float(self.param_groups[0]["lr"])
            with torch.cuda.device(tr.device):
                st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                rc = lib.hd_adam_step(C.c_void_p(tr.flat.data_ptr()), C.c_void_p(tr.grads.data_ptr()), C.c_void_p(m.data_ptr()),
                                      C.c_void_p(v.data_ptr()), tr.flat.numel(), lr, self.betas[0], self.betas[1], self.eps, k, scale, st)
            if rc != 0:
                raise L.HdError(rc, "hd_adam_step failed")
            self.state[tr.model] = (m, v, k)
            tr.weights_changed()
