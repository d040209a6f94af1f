#!/usr/bin/env python3
"""HBM roofline of the tile producer / stitcher on one GPU.

    python tools/bench_tiles.py [--n 24896] [--piece 64] [--res 10000] [--reps 20]

Algorithmic bytes: split reads and writes every tile element once (2 * ntiles * piece^2 * 4 B); stitch reads every
tile once and writes the whole n x n matrix (ntiles * piece^2 * 4 + n^2 * 4 B).  Timed with HIP events on the
launch stream, inputs resident in HBM; a bounded CPU leg (the oracle = numpy slicing, as the reference does) beside it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=24896)      # chr1 at 10 kb
    ap.add_argument("--piece", type=int, default=64)
    ap.add_argument("--res", type=int, default=10000)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    from hicdiff_amd import processdata as PD
    from oracle import tiles as OT
    dev = torch.device("cuda:0")
    m = torch.rand((a.n, a.n), device=dev)
    m = (m + m.T) / 2
    tiles, org = PD.split_pieces_device(m, a.piece, a.piece, a.res)
    back = PD.stitch_pieces_device(tiles, org, a.n)
    torch.cuda.synchronize()

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); torch.cuda.synchronize()
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps

    nt, pp = len(org), a.piece * a.piece
    # kernels only: the origin / lookup tables are built and uploaded once (they depend on n, piece, res alone)
    import ctypes as C
    from hicdiff_amd import _lib as L
    from hicdiff_amd.processdata.PrepareData_linear_sing import stitch_table
    lib = L.load()
    o_dev = torch.from_numpy(org.astype(np.int32)).to(dev)
    tb = stitch_table(org, a.n, a.piece)
    t_dev = torch.from_numpy(tb).to(dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    P = lambda x: C.c_void_p(x.data_ptr())
    ms_split = timed(lambda: lib.hd_split_pieces(P(m), a.n, P(o_dev), len(org), a.piece, P(tiles), st))
    ms_stitch = timed(lambda: lib.hd_stitch_pieces(P(tiles), P(t_dev), tb.shape[0], a.piece, a.piece, P(back), a.n, st))
    ms_split_api = timed(lambda: PD.split_pieces_device(m, a.piece, a.piece, a.res))
    b_split, b_stitch = 2 * nt * pp * 4, nt * pp * 4 + a.n * a.n * 4
    ncpu = min(a.n, 6000)
    mc = m[:ncpu, :ncpu].cpu().numpy()
    t0 = time.perf_counter(); ref = OT.split_pieces(mc, a.piece, a.piece, a.res); t_cpu = time.perf_counter() - t0
    print(json.dumps({
        "workload": f"n={a.n} piece={a.piece} res={a.res}: {nt} tiles", "dtype": "f32",
        "split": {"ms": round(ms_split, 4), "tiles_per_s": round(nt / ms_split * 1e3), "GBps": round(b_split / ms_split / 1e6, 1),
                  "frac_hbm": round(b_split / ms_split / 1e6 / 8000, 3), "ms_with_host_tables": round(ms_split_api, 4)},
        "stitch": {"ms": round(ms_stitch, 4), "GBps": round(b_stitch / ms_stitch / 1e6, 1), "frac_hbm": round(b_stitch / ms_stitch / 1e6 / 8000, 3)},
        "cpu_baseline": {"kind": "port", "cores": 1, "sample": f"split of the leading {ncpu}x{ncpu} block: {len(ref)} tiles",
                         "tiles_per_s": round(len(ref) / t_cpu)},
        "round_trip_exact": bool(torch.equal(PD.split_pieces_device(back, a.piece, a.piece, a.res)[0], tiles)),
    }))


if __name__ == "__main__":
    main()
