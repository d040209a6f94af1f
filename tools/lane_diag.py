#!/usr/bin/env python3
"""Diagnosis: is a difference between one whole-batch chain and two half-batch chains a batch dependence of some kernel or a fault of the
lane plumbing?  (a) eps of B tiles vs eps of its slices, stage by stage through the probes; (b) chains of 1, 2, 3, ... steps with
hd_set_chains 1 vs 2."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HICDIFF_GRAPHS", "1")
from _util import diffusion_class, product_unet, tiles  # noqa: E402

dim, mults, S, B = int(sys.argv[1]) if len(sys.argv) > 1 else 32, (1, 2, 4), 40, 6
m = product_unet("uncond", dim, mults)
eng = m.engine(torch.device("cuda", 0))
lib = eng.lib
lib.hd_debug_capture.argtypes = [C.c_void_p, C.c_int]
lib.hd_debug_read.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int32 * 4)]
x = tiles(3, B, S).cuda()
t = torch.full((B,), 17, device="cuda")


def probes(xs, ts):
    lib.hd_debug_capture(eng.ctx, 1)
    out = m(xs, ts)
    torch.cuda.synchronize()
    got = {"eps": out}
    for label in ("downs.2.1.out", "downs.2.1.h1", "downs.2.1.A1", "downs.2.1.B1", "downs.2.1.h2", "downs.2.1.A2", "downs.2.1.B2", "downs.2.2.ln_stats", "downs.2.2.q", "downs.2.2.ctx", "downs.2.2.att", "downs.2.2.y", "init_conv", "downs.0.0", "downs.0.2", "downs.0", "downs.1.0", "downs.1.2", "downs.1", "downs.2.0", "downs.2.2", "downs.2", "mid_attn", "mid",
                  "ups.0", "ups.1", "ups.2", "final_res"):
        dims = (C.c_int32 * 4)()
        if lib.hd_debug_read(eng.ctx, label.encode(), None, 0, C.byref(dims)) != 0:
            continue
        buf = torch.empty(tuple(dims), device="cuda")
        lib.hd_debug_read(eng.ctx, label.encode(), C.c_void_p(buf.data_ptr()), buf.numel(), C.byref(dims))
        got[label] = buf
    lib.hd_debug_capture(eng.ctx, 0)
    return got


full, a, b = probes(x, t), probes(x[:4], t[:4]), probes(x[4:], t[4:])
for k in full:
    part = torch.cat([a[k], b[k]])
    per = (full[k] - part).abs().flatten(1).max(dim=1).values
    print(f"(a) {k:20s} max|full - slices| = {float(per.max()):.3e}   per sample: " + " ".join(f"{float(v):.1e}" for v in per))

k = "downs.2.2.ln_stats"
fs, ps = full[k].reshape(B, -1, 2), torch.cat([a[k], b[k]]).reshape(B, -1, 2)
a2 = full["downs.2.1.out"].double()                      # (B, H, W, C)
a2s = torch.cat([a["downs.2.1.out"], b["downs.2.1.out"]])
print("a2 equal:", torch.equal(full["downs.2.1.out"], a2s))
mean = a2.mean(dim=-1).reshape(B, -1)
rstd = (a2.var(dim=-1, unbiased=False) + 1e-5).rsqrt().reshape(B, -1)
for nm, idx, ref in (("mean", 0, mean), ("rstd", 1, rstd)):
    df, dp = (fs[..., idx].double() - ref).abs(), (ps[..., idx].double() - ref).abs()
    nd = (fs[..., idx] != ps[..., idx]).sum(dim=1)
    print(nm, "pixels that differ per sample:", nd.tolist(), " err vs fp64: full", float(df.max()), "slices", float(dp.max()))
d = diffusion_class("uncond")(m, image_size=S, timesteps=30, loss_type="l2", beta_schedule="linear").cuda()
start = eng.randn(B, S, 5, 0, 30)
for n in (1, 2, 3, 4, 6):
    outs = []
    for ch in (1, 2):
        eng.set_chains(ch)
        xx = start.clone()
        with eng.chain(B, S):
            for tt in range(29, 29 - n, -1):
                d._step_inplace(xx, tt, None, eng=eng)
        torch.cuda.synchronize()
        outs.append(xx.clone())
    print(f"(b) {n} step(s): max|one chain - two chains| = {float((outs[0] - outs[1]).abs().max()):.3e}")
