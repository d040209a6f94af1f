#!/usr/bin/env python3
"""A/B table of the convolution kernel over the UNet's layer shapes (one process, one library):
    python tools/conv_suite.py [--lib _ra0] [--B 256] [--reps 5] [--only 64x64]
Each shape goes through the test-only hd_debug_conv entry; times are HIP events around the launch (hd_profile_*), the median of
--reps launches after one warm-up.  --lib TAG loads hicdiff_amd/libhicdiff_hip<TAG>.so (an experiment build: make TAG=... EXTRA=...)."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# name, S, C0, C1, Cout, K, mode bits (32 split-bf16, 64 = 32-channel activation slices, 8 = affine+SiLU loader, 1 = nearest x2 upsample)
SHAPES = [
    ("64->64 @64 plain ck16", 64, 64, 0, 64, 3, 32),
    ("64->64 @64 plain ck32", 64, 64, 0, 64, 3, 32 | 64),
    ("64->64 @64 affine ck32", 64, 64, 0, 64, 3, 32 | 64 | 8),
    ("64+64->64 @64 plain ck16", 64, 64, 64, 64, 3, 32),
    ("64+64->64 @64 plain ck32", 64, 64, 64, 64, 3, 32 | 64),
    ("128->128 @32 plain", 32, 128, 0, 128, 3, 32 | 64),
    ("128->128 @32 affine", 32, 128, 0, 128, 3, 32 | 64 | 8),
    ("128+128->128 @32 plain", 32, 128, 128, 128, 3, 32 | 64),
    ("256->256 @16 plain", 16, 256, 0, 256, 3, 32 | 64),
    ("256->256 @16 affine", 16, 256, 0, 256, 3, 32 | 64 | 8),
    ("512->512 @8 plain", 8, 512, 0, 512, 3, 32 | 64),
    ("512->512 @8 affine", 8, 512, 0, 512, 3, 32 | 64 | 8),
    ("128->64 @64 up x2", 32, 128, 0, 64, 3, 32 | 64 | 1),
    ("1x1 64+64->64 @64", 64, 64, 64, 64, 1, 32 | 64),
    ("1x1 128->128 @32", 32, 128, 0, 128, 1, 32 | 64),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="")
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--f16w2", action="store_true", help="3x3 shapes with two fp16 products per multiply (mode bit 512)")
    ap.add_argument("--f16w1", action="store_true", help="3x3 shapes with ONE fp16 product per multiply (mode bits 512 | 1024)")
    ap.add_argument("--zeros", action="store_true", help="all-zero activations and weights: the same instruction stream at the clock the chip holds on trivial data (DVFS check)")
    a = ap.parse_args()
    from hicdiff_amd import _lib as L
    if a.lib:
        L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), f"libhicdiff_hip{a.lib}.so")
    lib = L.load()
    lib.hd_debug_conv.restype = C.c_int
    lib.hd_debug_conv.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(1)
    total = 0.0
    for name, S, C0, C1, Cout, K, mode in SHAPES:
        if a.only and a.only not in name:
            continue
        B = a.B
        if (a.f16w2 or a.f16w1) and K == 3:
            mode |= 512 | (1024 if a.f16w1 else 0)
        x0 = torch.randn((B, S, S, C0), device="cuda", generator=g)
        x1 = torch.randn((B, S, S, C1), device="cuda", generator=g) if C1 else None
        cin = C0 + C1
        w = torch.randn((Cout, cin, K, K), device="cuda", generator=g) / (K * cin ** 0.5)
        b = torch.randn((Cout,), device="cuda", generator=g)
        A = torch.randn((B, cin), device="cuda", generator=g) * 0.5 + 1
        Bv = torch.randn((B, cin), device="cuda", generator=g)
        if a.zeros:
            x0.zero_(); w.zero_(); b.zero_()
            if x1 is not None:
                x1.zero_()
        So = S * 2 if mode & 1 else S
        out = torch.empty((B, So, So, Cout), device="cuda")
        aff = bool(mode & 8)
        call = lambda: lib.hd_debug_conv(p(x0), C0, p(x1), C1, B, S, S, p(w), p(b), Cout, K, mode, p(A) if aff else None, p(Bv) if aff else None,
                                         None, p(out), st)
        assert call() == 0, lib.hd_last_error(None)
        times = []
        for _ in range(a.reps):
            lib.hd_profile_enable(1)
            assert call() == 0
            torch.cuda.synchronize()
            rows = (L.HdProfileRow * L.HD_PROFILE_MAX_ROWS)()
            n = lib.hd_profile_read(rows, L.HD_PROFILE_MAX_ROWS)
            r = max((rows[i] for i in range(n)), key=lambda r: r.total_ms)
            times.append((r.total_ms * 1e3, r.kernel.decode(), r.flops))
            lib.hd_profile_enable(0)
        times.sort()
        us, kern, fl = times[len(times) // 2]
        total += us
        print(f"{name:28s} {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s-eq  min {times[0][0]:8.1f}  {kern.replace('conv_igemm_bf16x3_kernel', 'k').replace('conv_igemm_f16w2_kernel', 'kf16').replace('conv_igemm_f16w1_kernel', 'kf16x1')}")
    print(f"{'sum':28s} {total:8.1f} us   [lib '{a.lib or 'product'}', B={a.B}]")


if __name__ == "__main__":
    main()
