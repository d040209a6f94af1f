#!/usr/bin/env python3
"""Time one convolution shape through the test-only hd_debug_conv entry (kernel under study in isolation):
    python tools/conv_probe.py --B 64 --S 64 --cin 256 --cout 256 [--mode 288] [--reps 5] [--affine]
mode bits as in include/hicdiff_hip_debug.h (32 split-bf16 x3, 64 32-channel slices, 256 Winograd image, 8 affine+SiLU loader).
Prints the kernel time measured with HIP events around the launch (hd_profile_*)."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--S", type=int, default=64)
    ap.add_argument("--cin", type=int, default=64)
    ap.add_argument("--cout", type=int, default=64)
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--mode", type=int, default=32 | 256)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--affine", action="store_true")
    a = ap.parse_args()
    from hicdiff_amd import _lib as L
    lib = L.load()
    lib.hd_debug_conv.restype = C.c_int
    lib.hd_debug_conv.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((a.B, a.S, a.S, a.cin), device="cuda", generator=g)
    w = torch.randn((a.cout, a.cin, a.k, a.k), device="cuda", generator=g) / (a.k * a.cin ** 0.5)
    b = torch.randn((a.cout,), device="cuda", generator=g)
    A = torch.randn((a.B, a.cin), device="cuda", generator=g) * 0.5 + 1
    Bv = torch.randn((a.B, a.cin), device="cuda", generator=g)
    out = torch.empty((a.B, a.S, a.S, a.cout), device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    mode = a.mode | (8 if a.affine else 0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: lib.hd_debug_conv(p(x), a.cin, None, 0, a.B, a.S, a.S, p(w), p(b), a.cout, a.k, mode, p(A) if a.affine else None,
                                     p(Bv) if a.affine else None, None, p(out), st)
    assert call() == 0, lib.hd_last_error(None)
    lib.hd_profile_enable(1)
    for _ in range(a.reps):
        assert call() == 0
    torch.cuda.synchronize()
    rows = (L.HdProfileRow * L.HD_PROFILE_MAX_ROWS)()
    n = lib.hd_profile_read(rows, L.HD_PROFILE_MAX_ROWS)
    for i in range(n):
        r = rows[i]
        us = r.total_ms / r.launches * 1e3
        print(f"{r.kernel.decode():50s} {us:9.1f} us  {r.flops / r.total_ms / 1e9:7.1f} TFLOP/s-eq  {r.bytes / r.total_ms / 1e6:6.0f} GB/s  "
              f"[B={a.B} S={a.S} {a.cin}->{a.cout} mode={mode} ablate={os.environ.get('HICDIFF_ABLATE', '0')}]")
    lib.hd_profile_enable(0)
    if hasattr(lib, "hd_debug_conv_stamps") and not (a.mode & 256):     # make EXTRA=-DHD_STAMPS: per-workgroup cycle stamps of the implicit-GEMM kernel
        import numpy as np
        nwg = 4096
        buf = (C.c_ulonglong * (16 * nwg))()
        if lib.hd_debug_conv_stamps(buf, nwg, 1) == 0:
            sa = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 16).astype(np.int64)
            sa = sa[sa[:, 3] > 0]
            if len(sa):
                pro, loop, epi, bar = sa[:, 1] - sa[:, 0], sa[:, 2] - sa[:, 1], sa[:, 3] - sa[:, 2], sa[:, 4]
                print(f"  stamps over {len(sa)} workgroups (cycles of wave 0, mean / p10 / p90): prologue {pro.mean():.0f} / {np.percentile(pro, 10):.0f} / {np.percentile(pro, 90):.0f}"
                      f"   loop {loop.mean():.0f} / {np.percentile(loop, 10):.0f} / {np.percentile(loop, 90):.0f} (of which in barriers {bar.mean():.0f})"
                      f"   epilogue {epi.mean():.0f} / {np.percentile(epi, 10):.0f} / {np.percentile(epi, 90):.0f}")
                print(f"  inside the prologue (mean cycles): entry -> first window's loads issued {sa[:, 6].mean():.0f}, until they arrive {sa[:, 7].mean():.0f}"
                      f"   [entry -> tile decoded {sa[:, 8].mean():.0f}, tables written {sa[:, 9].mean():.0f}, first barrier {sa[:, 10].mean():.0f}, items set up {sa[:, 11].mean():.0f}]")
                print(f"  inside the epilogue (mean cycles): barriers {sa[:, 12].mean():.0f}, accumulators -> LDS {sa[:, 13].mean():.0f}, row passes + stores {sa[:, 14].mean():.0f}")
                # timeline of a few CUs (a CU's workgroups share a cycle counter; HW_ID bits 8-15 = CU / SH / SE): start, loop entry,
                # epilogue entry and end of each of its workgroups, in thousands of cycles from the CU's first start
                key = ((sa[:, 5] >> 32) << 8) | ((sa[:, 5] & 0xffff) >> 8)
                for k in np.unique(key)[:3]:
                    c = sa[key == k]
                    c = c[np.argsort(c[:, 0])]
                    t0 = c[0, 0]
                    print(f"  CU {int(k):#x}: {len(c)} workgroups; " + "  ".join(f"[{(r[0]-t0)/1e3:.0f} {(r[1]-t0)/1e3:.0f} {(r[2]-t0)/1e3:.0f} {(r[3]-t0)/1e3:.0f}]" for r in c))
    if hasattr(lib, "hd_debug_wino_stamps"):           # DIAG builds: in-kernel cycle stamps of one workgroup (conv_winograd.hip)
        buf = (C.c_ulonglong * 16)()
        if lib.hd_debug_wino_stamps(buf) == 0:
            for g in range(2):
                v = [buf[g * 8 + i] for i in range(6)]
                print(f"  stamps group {g}: T part {v[0]}  M part {v[1]}  wait B1 {v[2]}  wait Bmid {v[3]}  loop {v[4]}  epilogue-1 {v[5]}  (cycles, one wave)")


if __name__ == "__main__":
    main()
