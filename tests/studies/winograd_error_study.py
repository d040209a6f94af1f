"""CPU error study for Winograd F(2x2,3x3) with split-bf16 x3 products.

Question: if the stride-1 3x3 convolutions of the UNet are evaluated as
    Y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A
with the two transforms in fp32 (filter transform in fp64 at pack time), the 16 per-position channel
contractions in split-bf16 x3 (hi*hi + hi*lo + lo*hi, fp32 accumulate) -- 2.25x fewer MFMAs than the direct
form -- does a 50-step ancestral chain stay inside the 1e-3 parity bound against the fp32 oracle?

Run:  python tests/studies/winograd_error_study.py [--steps 50] [--size 64]
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
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def split(x):
    hi = x.to(torch.bfloat16).float()
    lo = (x - hi).to(torch.bfloat16).float()
    return hi, lo


def prod3(a_hi, a_lo, b_hi, b_lo, fn):
    return fn(a_hi, b_hi) + fn(a_hi, b_lo) + fn(a_lo, b_hi)


def direct_bf16x3(x, w, b):
    xh, xl = split(x)
    wh, wl = split(w)
    y = prod3(xh, xl, wh, wl, lambda a, c: _conv2d(a, c, None, padding=1))
    return y if b is None else y + b.view(1, -1, 1, 1)


def winograd(x, w, b, mode):
    """mode 'f32': fp32 products (isolates the transforms' own rounding); 'bf16x3': split products."""
    Bn, C, H, W = x.shape
    assert H % 2 == 0 and W % 2 == 0
    U = (G @ w.double() @ G.T).float()                                   # [Co, Ci, 4, 4]
    xp = F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                               # [B, C, H/2, W/2, 4, 4]
    V = BT @ d @ BT.T                                                    # fp32 adds
    if mode == "f32":
        M = torch.einsum("bcijxy,ocxy->boijxy", V, U)
    else:
        Vh, Vl = split(V)
        Uh, Ul = split(U)
        M = prod3(Vh, Vl, Uh, Ul, lambda a, c: torch.einsum("bcijxy,ocxy->boijxy", a, c))
    Y = AT @ M @ AT.T                                                    # [B, O, H/2, W/2, 2, 2]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(Bn, -1, H, W)
    return y if b is None else y + b.view(1, -1, 1, 1)


class Patch:
    """Routes the wide stride-1 3x3 convolutions of oracle.nets through `impl`; everything else stays fp32."""

    def __init__(self, impl, min_hw=8):
        self.impl, self.min_hw, self.hits = impl, min_hw, 0

    def __call__(self, x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        ok = (w.shape[-2:] == (3, 3) and stride in (1, (1, 1)) and padding in (1, (1, 1)) and w.shape[1] >= 64 and
              x.shape[-1] % 2 == 0 and x.shape[-1] >= self.min_hw)
        if not ok:
            return _conv2d(x, w, b, stride, padding, dilation, groups)
        self.hits += 1
        return self.impl(x, w, b)


def chain(model, S, T, seed, B):
    from oracle import diffusion as OD
    ref = OD.DiffusionRef(model, image_size=S, timesteps=T, beta_schedule="linear", loss_type="l2")
    return ref.p_sample_loop((B, 1, S, S), OD.TorchNoise(seed), keep_every=10)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=1)
    args = ap.parse_args()
    from _util import oracle_unet, tiles
    import oracle.nets as ON
    model = oracle_unet("uncond")
    S, T, B = args.size, args.steps, args.batch

    x = tiles(3, B, S)
    t = torch.tensor([500] * B)
    base = model(x, t)
    variants = {
        "direct bf16x3": direct_bf16x3,
        "winograd fp32 products": lambda a, w, b: winograd(a, w, b, "f32"),
        "winograd bf16x3 products": lambda a, w, b: winograd(a, w, b, "bf16x3"),
    }
    rel = lambda ref, got: ((ref - got).abs().max() / ref.abs().max()).item()
    print(f"UNet(64,(1,2,4,8)) at {S}x{S}, {B} tile(s); relative error = max|d| / max|ref| against the fp32 oracle")
    for name, impl in variants.items():
        p = Patch(impl)
        ON.F.conv2d = p
        try:
            got = model(x, t)
        finally:
            ON.F.conv2d = _conv2d
        print(f"  one forward   {name:28s} {rel(base, got):.2e}   ({p.hits} convolutions rerouted)")
    t0 = time.time()
    want, kept = chain(model, S, T, 11, B)
    print(f"  fp32 chain of {T} steps: {time.time() - t0:.0f} s")
    for name, impl in variants.items():
        ON.F.conv2d = Patch(impl)
        try:
            t0 = time.time()
            got, k2 = chain(model, S, T, 11, B)
        finally:
            ON.F.conv2d = _conv2d
        worst = max(rel(kept[k], k2[k]) for k in kept)
        print(f"  {T}-step chain {name:28s} final {rel(want, got):.2e}  worst kept state {worst:.2e}   ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
