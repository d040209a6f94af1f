#!/usr/bin/env python3
"""Time the direct weight-gradient kernel on one shape (GPU box; `make EXTRA=-DHD_STAMPS` adds its in-kernel phase stamps):
    python3 tools/wgrad_probe.py --B 64 --S 64 --cin 256 --cout 256 --k 3"""
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
    ap.add_argument("--cin", type=int, default=256)
    ap.add_argument("--cout", type=int, default=256)
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    from hicdiff_amd import _lib as L
    lib = L.load()
    P = C.c_void_p
    fn = lib.hd_debug_conv_wgrad_direct
    fn.restype = C.c_int
    fn.argtypes = [P, C.c_int, P, C.c_int, P] + [C.c_int] * 5 + [P, P, P, P, P, C.c_int, P]
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((a.B, a.S, a.S, a.cin), device="cuda", generator=g)
    gr = torch.randn((a.B, a.S, a.S, a.cout), device="cuda", generator=g)
    out = torch.empty((a.cout, a.cin, a.k, a.k), device="cuda")
    db = torch.empty(a.cout, device="cuda")
    st = P(torch.cuda.current_stream().cuda_stream)
    call = lambda: fn(P(x.data_ptr()), a.cin, P(), 0, P(gr.data_ptr()), a.B, a.S, a.S, a.cout, a.k, P(out.data_ptr()), P(db.data_ptr()), P(), P(), P(), 0, st)
    assert call() == 0
    lib.hd_profile_enable(1)
    for _ in range(a.reps):
        assert call() == 0
    torch.cuda.synchronize()
    rows = (L.HdProfileRow * L.HD_PROFILE_MAX_ROWS)()
    n = lib.hd_profile_read(rows, L.HD_PROFILE_MAX_ROWS)
    for i in range(n):
        r = rows[i]
        us = r.total_ms / r.launches * 1e3
        print(f"{r.kernel.decode():40s} {us:9.1f} us  {r.flops / r.total_ms / 1e9:7.1f} TFLOP/s-eq  [B={a.B} S={a.S} {a.cin}->{a.cout} k={a.k}]")
    lib.hd_profile_enable(0)
    if hasattr(lib, "hd_debug_wgd_stamps"):
        import numpy as np
        buf = (C.c_ulonglong * (5 * 1024))()
        if lib.hd_debug_wgd_stamps(buf, 1024) == 0:
            s = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 5).astype(np.int64)
            s = s[s[:, 3] > 0]
            print(f"  stamps over {len(s)} workgroups (cycles of wave 0, mean): store phase incl. wait for the prefetch {s[:, 0].mean():.0f}, barriers {s[:, 1].mean():.0f}, "
                  f"request + MFMA phase {s[:, 2].mean():.0f} (nine-tap kernel: MFMA phase alone; its request phase {s[:, 4].mean():.0f}), total {s[:, 3].mean():.0f}")


if __name__ == "__main__":
    main()
