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
        self._loaded = None        # a load_state_dict() waiting for the flat buffers of the first native step

    # ---- torch.optim.Adam's checkpoint form (pretrain/train_hicedrn_Diff.py:93-96 saves {'epoch', 'model_state_dict', 'optimizer_state_dict'})
    def state_dict(self):
        """The layout torch.optim.Adam.state_dict() has: per-parameter 'step' / 'exp_avg' / 'exp_avg_sq' keyed by the parameter's index, cut out of
        the flat moment buffers; loads into a torch.optim.Adam over the same parameters, and back."""
        state = {}
        for i, p in enumerate(self.params):
            tr, off = getattr(p, "_hd_flat", (None, 0))
            ent = self.state.get(tr.model) if tr is not None else None
            if ent is None and self._loaded is not None and i in self._loaded["state"]:
                state[i] = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in self._loaded["state"][i].items()}
            if ent is None:
                continue
            m, v, k = ent
            n = p.numel()
            state[i] = {"step": torch.tensor(float(k)), "exp_avg": m[off:off + n].view(p.shape).clone(), "exp_avg_sq": v[off:off + n].view(p.shape).clone()}
        group = {"lr": float(self.param_groups[0]["lr"]), "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.params):
            raise ValueError("loaded state dict has a different number of parameter groups / parameters")
        g = groups[0]
        if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False):
            raise NotImplementedError("the reference trains with plain Adam (train.py:111)")
        self.lr, self.betas, self.eps = float(g["lr"]), (float(g["betas"][0]), float(g["betas"][1])), float(g["eps"])
        self.param_groups[0].update(lr=self.lr, betas=self.betas, eps=self.eps)
        order = {pid: i for i, pid in enumerate(g["params"])}
        self._loaded = {"state": {order[pid]: st for pid, st in sd["state"].items() if pid in order}}
        self.state = {}            # the moments are rebuilt from the loaded tensors at the next step
        steps = {int(float(st["step"])) for st in self._loaded["state"].values()}
        if len(steps) > 1:
            raise NotImplementedError("per-parameter step counts differ: the flat Adam kernel keeps one step count per network")

    def _moments(self, tr):
        """(m, v, step) of a trainer's flat buffer: kept from the last step, or rebuilt from a loaded state dict, or zeros."""
        m, v, k = self.state.get(tr.model, (None, None, 0))
        if m is None or m.numel() != tr.flat.numel() or m.device != tr.flat.device:
            m, v = torch.zeros_like(tr.flat), torch.zeros_like(tr.flat)
            if self._loaded is not None:
                index = {id(p): i for i, p in enumerate(self.params)}
                for p in tr.params:
                    st = self._loaded["state"].get(index.get(id(p), -1))
                    if st is None:
                        continue
                    off, n = p._hd_flat[1], p.numel()
                    m[off:off + n].copy_(st["exp_avg"].reshape(-1))
                    v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                    k = int(float(st["step"]))
                    del self._loaded["state"][index[id(p)]]           # consumed: the flat buffers carry it from here on
                if not self._loaded["state"]:
                    self._loaded = None
        return m, v, k

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
            m, v, k = self._moments(tr)
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
