"""Diagnostic: per-phase cycle SHARES of the split-bf16 convolution main loop (build the library with
`make -C hicdiff_amd/csrc clean all STAMP=1` first; never quote run times of that build).

    python tools/conv_stamp_report.py
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_kernels import rnd, run_conv  # noqa: E402
from hicdiff_amd import _lib as L  # noqa: E402

lib = L.load()
lib.hd_debug_stamp.argtypes = [C.c_int, C.c_ulonglong * 8]


def measure(name, fn):
    fn()
    out = (C.c_ulonglong * 8)()
    lib.hd_debug_stamp(1, out)
    fn()
    lib.hd_debug_stamp(2, out)
    w, b, m, s, ep, tot, waves, nit = [float(v) for v in out]
    if waves == 0:
        print("library was not built with STAMP=1")
        return
    sh = lambda v: 100.0 * v / tot
    print(f"{name:32s} weight-stage {sh(w):5.1f}%  barrier {sh(b):5.1f}%  reads+MFMA {sh(m):5.1f}%  slice-stage {sh(s):5.1f}%  "
          f"epilogue {sh(ep):5.1f}%  other {sh(tot - w - b - m - s - ep):5.1f}%   (iterations/wave {nit / waves:.0f})")


for (B, S, Cin, Cout) in [(64, 64, 256, 256), (64, 64, 64, 64), (64, 32, 128, 128), (64, 8, 512, 512)]:
    x, w, b = rnd(1, B, Cin, S, S), rnd(2, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), rnd(3, Cout)
    measure(f"3x3 B{B} S{S} {Cin}->{Cout}", lambda: run_conv(x, None, w, b, 3, 96))
