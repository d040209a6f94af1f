"""CPU error study: can the wide convolutions run on TWO matrix-core products per multiply instead of three?  

The product kernels evaluate x*w as split-bf16 x3 (xh wh + xh wl + xl wh, 2^-16-ish per product) because one bf16 product
(2^-9 per operand) breaks the 1e-3 parity bound over a sampling chain.  The convolution kernels are power-bound at three
MFMAs per product (section 4c), so the only way to make them faster is fewer MFMAs.  fp16 has 11 significand bits against bf16's 8 and the
same MFMA rate, which opens two-product forms:
    f16 a2w1:  (xh + xl) wh        activations exact to 2^-22, weights rounded to fp16 (2^-12 relative, fixed per weight)
    f16 a1w2:  xh (wh + wl)        weights exact, activations rounded to fp16 (fresh rounding every step)
    f16 x1:    xh wh               one product
and for reference bf16 x1, bf16 a2w1 and the product's bf16 x3.  fp16 subnormals (|v| < 6.1e-5) are kept or flushed (--flush).
Question: per-forward and 50-step-chain error against the fp32 oracle, bound 1e-3 (max|d| / max|ref|).

Run:  python tests/studies/two_product_error_study.py [--steps 50] [--size 64] [--flush]
Test infrastructure (imports oracle/); nothing here is on the product path.
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

_conv2d = F.conv2d
FLUSH = False


def rnd(x, dt):
    y = x.to(dt).float()
    if FLUSH and dt == torch.float16:
        y = torch.where(y.abs() < 6.103515625e-05, torch.zeros_like(y), y)
    return y


def split(x, dt):
    hi = rnd(x, dt)
    return hi, rnd(x - hi, dt)


def make(dt, a_terms, w_terms):
    def impl(x, w, b, stride, padding):
        conv = lambda a, c: _conv2d(a, c, None, stride, padding)
        xh, xl = split(x, dt)
        wh, wl = split(w, dt)
        y = conv(xh, wh)
        if a_terms == 2:
            y = y + conv(xl, wh)
        if w_terms == 2:
            y = y + conv(xh, wl)
        return y if b is None else y + b.view(1, -1, 1, 1)
    return impl


class Patch:
    """Routes every convolution with at least 64 input channels (3x3 and 1x1: the layers the split-bf16 kernels take) through `impl`."""

    def __init__(self, impl):
        self.impl, self.hits = impl, 0

    def __call__(self, x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        if w.shape[1] < 64 or groups != 1:
            return _conv2d(x, w, b, stride, padding, dilation, groups)
        self.hits += 1
        return self.impl(x, w, b, stride, padding)


def chain(model, S, T, seed, B):
    from oracle import diffusion as OD
    ref = OD.DiffusionRef(model, image_size=S, timesteps=T, beta_schedule="linear", loss_type="l2")
    return ref.p_sample_loop((B, 1, S, S), OD.TorchNoise(seed), keep_every=10)


def main():
    global FLUSH
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--flush", action="store_true", help="flush fp16 subnormals to zero (operands)")
    ap.add_argument("--net", default="unet", choices=["unet", "hicedrn"])
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    FLUSH = args.flush
    from _util import oracle_unet, tiles
    import oracle.nets as ON
    if args.net == "unet":
        model = oracle_unet("uncond")
    else:
        from _util import oracle_hicedrn
        model = oracle_hicedrn("uncond", 8)
    S, T, B = args.size, args.steps, args.batch
    x = tiles(3, B, S)
    t = torch.tensor([500] * B)
    base = model(x, t)
    variants = {
        "bf16 x3 (product)": make(torch.bfloat16, 2, 2),
        "bf16 x1": make(torch.bfloat16, 1, 1),
        "bf16 a2w1": make(torch.bfloat16, 2, 1),
        "f16 x1": make(torch.float16, 1, 1),
        "f16 a2w1": make(torch.float16, 2, 1),
        "f16 a1w2": make(torch.float16, 1, 2),
    }
    if args.only:
        variants = {k: v for k, v in variants.items() if any(o in k for o in args.only.split(","))}
    rel = lambda ref, got: ((ref - got).abs().max() / ref.abs().max()).item()
    print(f"{args.net} at {S}x{S}, {B} tile(s), flush={FLUSH}; relative error = max|d| / max|ref| against the fp32 oracle", flush=True)
    for name, impl in variants.items():
        p = Patch(impl)
        ON.F.conv2d = p
        try:
            got = model(x, t)
        finally:
            ON.F.conv2d = _conv2d
        print(f"  one forward   {name:20s} {rel(base, got):.2e}   ({p.hits} convolutions rerouted)", flush=True)
    t0 = time.time()
    want, kept = chain(model, S, T, 11, B)
    print(f"  fp32 chain of {T} steps: {time.time() - t0:.0f} s", flush=True)
    for name, impl in variants.items():
        ON.F.conv2d = Patch(impl)
        try:
            t0 = time.time()
            got, k2 = chain(model, S, T, 11, B)
        finally:
            ON.F.conv2d = _conv2d
        worst = max(rel(kept[k], k2[k]) for k in kept)
        print(f"  {T}-step chain {name:20s} final {rel(want, got):.2e}  worst kept state {worst:.2e}   ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
