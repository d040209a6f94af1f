#!/usr/bin/env python3
"""Run the large-grid LayerNorm-loader 1x1 convolution (tests/test_gpu_kernels.py::test_large_grid_*[ln1x1]) against ANY build of the
library, also ones that predate the current ABI (no symbol checks): used by the zero-lane hazard bisect.
    python tools/ln_zero_lane_probe.py path/to/libhicdiff_hip*.so [...]
Prints, per library and run: relative error against a torch fp32 reference, number of exact zeros in the output, repeatability."""
import ctypes as C
import sys

import torch
import torch.nn.functional as F


def main():
    torch.manual_seed(0)
    import os
    B, S, Cin, Cout = int(os.environ.get('PB', 32)), int(os.environ.get('PS', 64)), int(os.environ.get('PCIN', 64)), int(os.environ.get('PCOUT', 384))
    g = torch.Generator().manual_seed(1)
    x = (torch.rand((B, Cin, S, S), generator=g) * 2 - 1) * 2 + 0.5
    w = (torch.rand((Cout, Cin, 1, 1), generator=g) * 2 - 1) / 8
    gain = (torch.rand((Cin,), generator=g) * 2 - 1) * 0.2 + 1
    xd, wd, gd = x.cuda(), w.cuda(), gain.cuda()
    var, mean = xd.var(dim=1, unbiased=False, keepdim=True), xd.mean(dim=1, keepdim=True)
    ref = F.conv2d((xd - mean) * (var + 1e-5).rsqrt() * gd.view(1, -1, 1, 1), wd).permute(0, 2, 3, 1).contiguous()
    xn = xd.permute(0, 2, 3, 1).contiguous()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    for path in sys.argv[1:]:
        lib = C.CDLL(path)
        lib.hd_debug_conv.restype = C.c_int
        lib.hd_debug_conv.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        outs = []
        for run in range(4):
            out = torch.full((B, S, S, Cout), float("nan"), device="cuda")
            rc = lib.hd_debug_conv(p(xn), Cin, None, 0, B, S, S, p(wd), None, Cout, 1, 16 | 32 | 64, p(gd), None, None, p(out), st)
            torch.cuda.synchronize()
            assert rc == 0, rc
            outs.append(out)
        err = [((o - ref).abs().max() / ref.abs().max()).item() for o in outs]
        zeros = [int((o == 0).sum().item()) for o in outs]
        bad = [int(((o - ref).abs() > 1e-3 * ref.abs().max()).sum().item()) for o in outs]
        same = all(torch.equal(outs[0], o) for o in outs[1:])
        if bad[0]:
            d = (outs[0] - ref).abs() > 1e-3 * ref.abs().max()
            pix = d.any(dim=3).reshape(B, -1)                         # [B][S*S] pixels with any bad channel
            idx = pix.nonzero()
            lin = (idx[:, 0] * S * S + idx[:, 1])
            print(f"    bad pixels {int(pix.sum())} of {B * S * S}; first linear pixel indices {lin[:12].tolist()}; pixel index mod 128 histogram (top) "
                  f"{torch.bincount(lin % 128, minlength=128).topk(6).indices.tolist()}; bad channels per bad pixel (mean) {float(d.sum()) / max(int(pix.sum()), 1):.1f}; "
                  f"samples hit {int(pix.any(dim=1).sum())}")
        if bad[0]:
            # does a bad row equal the CORRECT row of some other pixel of its sample (an addressing mix-up), or of the un-normalised input?
            o, r = outs[0].reshape(B, S * S, Cout), ref.reshape(B, S * S, Cout)
            bp = d.any(dim=3).reshape(B, -1).nonzero()[:6]
            for b_, p_ in bp.tolist():
                dist = (r[b_] - o[b_, p_]).abs().max(dim=1).values                   # distance to every correct row of the sample
                q = int(dist.argmin())
                print(f"    sample {b_} pixel {p_}: nearest correct row is pixel {q} (offset {q - p_}) at max|diff| {float(dist[q]):.2e}; own row diff {float(dist[p_]):.2e}; "
                      f"|out| mean {float(o[b_, p_].abs().mean()):.3f} vs |ref| mean {float(r[b_, p_].abs().mean()):.3f}; ratio out/ref on ch0..3 {[round(float(o[b_, p_, c] / r[b_, p_, c]), 3) for c in range(4)]}")
        print(f"{path.split('/')[-1]:32s} rel err {['%.1e' % e for e in err]}  exact zeros {zeros}  elements off by > 1e-3 {bad}  repeatable {same}", flush=True)


if __name__ == "__main__":
    main()
