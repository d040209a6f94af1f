"""CPU error study: an error BUDGET for two-product arithmetic -- which (timestep band) x (layer class) cells of a reverse chain could run
`xh (wh + wl)` on fp16 operands (two matrix-core products per multiply) instead of split-bf16 x3 and keep the chain inside the parity bound?
(follows tests/studies/two_product_error_study.py, which rerouted every wide convolution at every step and ended
3-8 x outside the bound.)

Why there could be room: the state after step t is  x_{t-1} = c1(t) clip(x0) + c2(t) x_t + sigma_t z  (src/hicdiff.py:584-601).  An error in
the network's output reaches x0 through sqrt_recipm1_alphas_cumprod[t] and x_{t-1} through c1(t) (1e-4 at t = 999 of 1000, growing to 1 at
t = 0), and where the clamp saturates it does not get through at all; the drift table of DESIGN section 2 shows the x3 arithmetic's own error
(1.4e-5 per forward) arriving as 6.6e-7 after 100 steps and 1e-4 at the end.

Arithmetic under test, per cell:   fp16 a1w2 = xh (wh + wl): weights exact to 2^-22, activations rounded to fp16 per use; everything else
(and every cell outside the policy) runs split-bf16 x3 as the product does.  Error = max|d| / max|ref| of the FINAL tiles against the fp32
oracle over the same noise; the bound for a cell to count is 5e-4 (half of the 1e-3 parity bound, leaving the other half to the kernels'
summation order).  Also reported: the share of the chain's wide-convolution MFMA instructions the policy moves to two products (a cell
saves a third of its own).

    python tests/studies/error_budget_study.py [--steps 50 --size 64] [--full]     (--full: the 1000-step chain at 40 x 40)
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
NBANDS = 4
CLASSES = ("full", "half", "low")          # feature map side: S, S/2, <= S/4


def split(x, dt):
    hi = x.to(dt).float()
    return hi, (x - hi).to(dt).float()


def conv_x3(x, w, stride, padding):
    xh, xl = split(x, torch.bfloat16)
    wh, wl = split(w, torch.bfloat16)
    c = lambda a, b: _conv2d(a, b, None, stride, padding)
    return c(xh, wh) + c(xh, wl) + c(xl, wh)


def conv_a1w2(x, w, stride, padding):
    xh = x.to(torch.float16).float()
    wh, wl = split(w, torch.float16)
    c = lambda a, b: _conv2d(a, b, None, stride, padding)
    return c(xh, wh) + c(xh, wl)


def conv_x1(x, w, stride, padding):
    """fp16 x fp16, one product: both operands rounded once."""
    return _conv2d(x.to(torch.float16).float(), w.to(torch.float16).float(), None, stride, padding)


class Router:
    """F.conv2d stand-in inside oracle.nets: wide convolutions (>= 64 input channels) go through x3 or, in the cells of `policy`
    (set of (band, class)), through fp16 a1w2.  Counts MFMA work (products x flops) per cell."""

    def __init__(self, S, T, policy, x1_cells=()):
        self.S, self.T, self.policy, self.t = S, T, policy, 0
        self.x1_cells = set(x1_cells)   # cells that take ONE fp16 product instead (the --x1 experiment)
        self.work = {}                  # (band, class) -> algorithmic flops of the wide convolutions

    def cell(self, x):
        side = x.shape[-1]
        cls = "full" if side >= self.S else "half" if side * 2 >= self.S else "low"
        return min(NBANDS - 1, self.t * NBANDS // self.T), cls

    def __call__(self, x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        if w.shape[1] < 64 or groups != 1:
            return _conv2d(x, w, b, stride, padding, dilation, groups)
        cell = self.cell(x)
        y = (conv_x1 if cell in self.x1_cells else conv_a1w2 if cell in self.policy else conv_x3)(x, w, stride, padding)
        self.work[cell] = self.work.get(cell, 0.0) + 2.0 * y.numel() * w.shape[1] * w.shape[2] * w.shape[3]
        return y if b is None else y + b.view(1, -1, 1, 1)


def run_chain(model, S, T, B, seed, router):
    from oracle import diffusion as OD
    import oracle.nets as ON

    def timed_model(x, t, cond=None):
        if router is not None:
            router.t = int(t.reshape(-1)[0])
        return model(x, t, cond)

    ref = OD.DiffusionRef(timed_model, image_size=S, timesteps=T, beta_schedule="linear", loss_type="l2")
    if router is not None:
        ON.F.conv2d = router
    try:
        return ref.p_sample_loop((B, 1, S, S), OD.TorchNoise(seed))
    finally:
        ON.F.conv2d = _conv2d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--full", action="store_true", help="T = 1000 at 40 x 40 (the drift test's chain); band / cumulative policies only")
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--x1", action="store_true", help="only the three-tier experiment: ONE fp16 product in band 3 (and in bands 3 + 2), two products in band 2")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    from _util import oracle_unet
    model = oracle_unet("uncond")
    S, T, B = (40, 1000, args.batch) if args.full else (args.size, args.steps, args.batch)
    rel = lambda ref, got: ((ref - got).abs().max() / ref.abs().max()).item()
    t0 = time.time()
    want = run_chain(model, S, T, B, 11, None)
    print(f"UNet(64,(1,2,4,8)) at {S}x{S}, {B} tile(s), T = {T}, linear schedule; fp32 oracle chain {time.time() - t0:.0f} s", flush=True)
    print("band b = timesteps [b T/4, (b+1) T/4): band 3 is the START of the reverse chain (t near T), band 0 its end", flush=True)

    bands, all_cls = range(NBANDS), set(CLASSES)
    policies = [("x3 everywhere (product)", set())]
    if not args.full:
        policies += [(f"band {b} x {c}", {(b, c)}) for b in reversed(bands) for c in CLASSES]
    policies += [(f"band {b}, all classes", {(b, c) for c in all_cls}) for b in reversed(bands)]
    policies += [(f"bands >= {b}, all classes", {(bb, c) for bb in bands if bb >= b for c in all_cls}) for b in (2, 1)]
    policies += [(f"class {c}, all bands", {(b, c) for b in bands}) for c in CLASSES]
    policies += [("everything (section 4d)", {(b, c) for b in bands for c in all_cls})]
    x1 = {}
    if args.x1:
        b3, b2 = {(3, c) for c in all_cls}, {(2, c) for c in all_cls}
        policies = [("x3 everywhere (product)", set()), ("band 3: x1, band 2: a1w2", b3 | b2), ("band 3: x1 only", b3), ("bands 3, 2: x1", b3 | b2)]
        x1 = {"band 3: x1, band 2: a1w2": b3, "band 3: x1 only": b3, "bands 3, 2: x1": b3 | b2}
    total = None
    for name, pol in policies:
        r = Router(S, T, pol, x1.get(name, ()))
        t0 = time.time()
        got = run_chain(model, S, T, B, 11, r)
        if total is None:
            total = sum(r.work.values())
            shares = {k: v / total for k, v in sorted(r.work.items())}
            print("  share of the chain's wide-convolution MFMA work per cell: " + ", ".join(f"{k}: {v:.3f}" for k, v in shares.items()), flush=True)
        moved = sum(v for k, v in r.work.items() if k in pol) / total
        e = rel(want, got)
        print(f"  {name:28s} final error {e:.2e}  {'ok ' if e < 5e-4 else 'OUT'}  cells' share of MFMA work {moved:.3f} -> {moved / 3:.3f} of the MFMA instructions saved"
              f"   ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
