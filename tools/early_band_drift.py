#!/usr/bin/env python3
"""Where may the early band end?  1000-step ancestral chains on the GPU (device noise, graph replay, the path bench.py times) with the 3x3
convolutions on two fp16 products per multiply for t >= f * T, against the CPU oracle over the same noise: final relative error
max|d| / max|ref| per (network flavour, tile size, seed, f).  f = 1: split-bf16 x3 at every step (the baseline).  Parity bound: 1e-3.

    python3 tools/early_band_drift.py [--seeds 2026 7 99] [--fracs 1 0.5 0.25 0]        (on a GPU box; minutes of CPU oracle per seed)
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
os.environ.setdefault("HICDIFF_GRAPHS", "1")
from _util import diffusion_class, oracle_unet, product_hicedrn, product_unet, rel_err, tiles  # noqa: E402
from test_gpu_timed_path import AncestralDeviceNoise  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[2026, 7, 99])
    ap.add_argument("--fracs", type=float, nargs="+", default=[1.0, 0.5, 0.25, 0.0])
    ap.add_argument("--cases", nargs="*", default=["uncond:40", "cond:64"])
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--schedule", default="linear", choices=["linear", "cosine", "sigmoid"], help="beta schedule of the chains (the reference's class default and inference.py: sigmoid; train.py and bench.py: linear)")
    ap.add_argument("--late-low", action="store_true", help="below the band: two fp16 products on the maps of at most (S/4)^2 pixels")
    ap.add_argument("--x1-from", type=float, default=0.75, help="inside the band, steps t >= this fraction of T take ONE fp16 product (2: never)")
    ap.add_argument("--vs-gpu", nargs="*", default=[], help="cases (e.g. hicedrn:40 hicedrn:64 uncond:64) measured against the GPU's own split-bf16 x3 chain "
                    "instead of the CPU oracle: what the band ADDS to that chain's error (the x3 chain's own error is what the drift tests bound)")
    a = ap.parse_args()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    from oracle import diffusion as OD
    T, B = 1000, a.batch
    vs_gpu(a)
    for case in a.cases:
        kind, S = case.split(":")[0], int(case.split(":")[1])
        net, ref_net = product_unet(kind), oracle_unet(kind)
        d = diffusion_class(kind)(net, image_size=S, timesteps=T, loss_type="l2", beta_schedule=a.schedule).cuda()
        d.early_band_x1_from, d.late_band_low_f16 = a.x1_from, a.late_low
        for seed in a.seeds:
            lq = tiles(seed, B, S) if kind != "uncond" else None
            ref = OD.DiffusionRef(ref_net, image_size=S, timesteps=T, beta_schedule=a.schedule, loss_type="l2", kind=kind)
            t0 = time.time()
            want = ref.p_sample_loop(lq if lq is not None else (B, 1, S, S), AncestralDeviceNoise(B, S, T, seed))
            line = f"{kind:6s} {S}x{S} seed {seed:5d} (oracle {time.time() - t0:4.0f} s):"
            for f in a.fracs:
                d.seed, d.early_band_from, d.early_band_f16 = seed, f, f < 1.0
                got = d.sample(torch.zeros(B, 1, S, S)) if kind == "uncond" else d.super_resolution(lq.cuda())
                line += f"  f={f:4.2f}: {rel_err(want, got):.2e}"
            print(line, flush=True)


def vs_gpu(a):
    T = 1000
    for case in a.vs_gpu:
        kind, S = case.split(":")[0], int(case.split(":")[1])
        net = product_hicedrn("uncond", 32) if kind == "hicedrn" else product_unet(kind)
        net.EARLY_BAND_OK = True
        flavour = "uncond" if kind == "hicedrn" else kind
        d = diffusion_class(flavour)(net, image_size=S, timesteps=T, loss_type="l2", beta_schedule=a.schedule).cuda()
        d.early_band_x1_from, d.late_band_low_f16 = a.x1_from, a.late_low
        B = a.batch
        for seed in a.seeds:
            lq = tiles(seed, B, S) if flavour != "uncond" else None
            run = (lambda: d.sample(torch.zeros(B, 1, S, S))) if flavour == "uncond" else (lambda: d.super_resolution(lq.cuda()))
            d.seed, d.early_band_f16 = seed, False
            want = run()
            line = f"{kind:7s} {S}x{S} seed {seed:5d} (reference: the GPU's x3 chain):"
            for f in a.fracs:
                if f >= 1.0:
                    continue
                d.early_band_from, d.early_band_f16 = f, True
                line += f"  f={f:4.2f}: {rel_err(want, run()):.2e}"
            print(line, flush=True)


if __name__ == "__main__":
    main()
