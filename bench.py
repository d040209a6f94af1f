#!/usr/bin/env python3
"""Headline benchmark: denoised Hi-C tiles / second for a 1000-step reverse (ancestral DDPM) chain.

    python bench.py [--gpus N --steps K --warmup W] [--workload unet64|unet40|hicedrn64|unet64cond|hicedrn64cond]
    python bench.py --workload hicedrn64_train [--batch 64 --steps 5 --warmup 2]     (native training step)
    python bench.py --workload tiles                                                (tile producer / stitcher, section 8 f-3)
    python bench.py --gpus N                       (N > 1 without RANK in the environment: starts the N rank processes itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --full-chain                   (time all 1000 steps of one chain; also implied by --steps >= 1000)
    python bench.py --gpus N --workload hicedrn64 --total-tiles 256      (strong-scaling form of BASELINE configs[3]: 256/N tiles per GPU)

A "step" is one pass of the hot path over one batch: the epsilon-network forward plus the fused
posterior update for B tiles (one hd_ddpm_step call).  Since round 4 a step's cost depends on the part of the
chain it sits in (the precision schedule of the 3x3 convolutions: one fp16 product for t >= 3T/4, two for
t >= T/2, two on the low-resolution maps only below that), so the K timed steps are spread
evenly over the chain (t = 999, 999 - 1000/K, ...): a stratified sample whose mean is the chain's mean when K is a
multiple of 4 (the default 20); tiles/s = tiles per batch / (1000 * mean seconds per step).
The line also carries `sustained`: a whole 1000-step chain (or as much of one as fits the time budget) timed
after the K-step region.  Tiles shard across ranks with no data-path collective (weak scaling: B tiles per
GPU); the one RCCL all-gather of the finished tiles is exercised after the timed region and reported
separately.  On each GPU the batch runs as two half-batch chains on two streams (hd_chain_begin / _end,
DESIGN.md section 7b; `--chains 1` for one).

Inputs are synthetic and already resident in HBM when the timed region starts: x_T from the
device Philox generator, random-init weights of the named architecture (seeded).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GOMP_SPINCOUNT", "20000")   # the cpu_baseline leg runs one OpenMP worker per granted core: bounded barrier spins (tests/conftest.py), set before torch loads libgomp

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic work per tile-step, SURVEY.md section 8(d) (forward hooks on the reference modules):
# every conv/linear reads its input once and writes its output once in fp32; weights once per batch-step.
WORK = {
    "unet64": dict(arch="unet", cond=False, S=64, B=256, flop=14.358e9, act_bytes=92.6e6, w_bytes=142.7e6),
    "unet64cond": dict(arch="unet", cond=True, S=64, B=256, flop=14.384e9, act_bytes=92.6e6, w_bytes=142.8e6),
    "unet40": dict(arch="unet", cond=False, S=40, B=64, flop=5.611e9, act_bytes=36.2e6, w_bytes=142.7e6),
    "hicedrn64": dict(arch="hicedrn", cond=False, S=64, B=256, flop=314.143e9, act_bytes=553.7e6, w_bytes=150.3e6),
    "hicedrn64cond": dict(arch="hicedrn", cond=True, S=64, B=256, flop=314.162e9, act_bytes=553.7e6, w_bytes=150.3e6),   # + the second input plane of the head
}
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA; the split-bf16 x3 conv issues 3 MFMA flops per algorithmic flop
PEAK_HBM_GBS = 8000.0
T_CHAIN = 1000
# HBM bytes per launch of each kernel from the FETCH_SIZE / WRITE_SIZE passes of this round (separate rocprofv3
# --pmc runs of this script; profiles/README.md): a recorded measurement, reported as roofline.traffic with its source.
TRAFFIC_FILE = {"unet64": os.path.join(ROOT, "profiles", "r04_s3_unet64_b256_hbm_traffic.json")}


def recorded_traffic(workload, kernel, batch, default_batch):
    path = TRAFFIC_FILE.get(workload)
    if batch != default_batch or not path or not os.path.exists(path):
        return None, None
    with open(path) as f:
        rec = json.load(f)
    for name, row in rec["kernels"].items():
        if kernel in name:
            return row["fetch_bytes_per_launch"] + row["write_bytes_per_launch"], os.path.relpath(path, ROOT)
    return None, None


def rank_env(args):
    """(rank, local device index, world, dist-or-None).  HICDIFF_BENCH_BACKEND=gloo and HICDIFF_DEVICE=<i> let several ranks
    rehearse on one GPU (tests); the default is one GPU per rank over RCCL."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    local = int(os.environ.get("HICDIFF_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("HICDIFF_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, device, world, dist


def max_over_ranks(seconds, dist, device):
    if dist is None:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def host_threads():
    """CPU-baseline threads.  BASELINE.md section 3 asks for the box's host cores; a GPU box SHOWS every core of its host (os.cpu_count() = 128
    or more) but grants one job a share of 16 per GPU, and torch on one thread per visible core crawls under that oversubscription (measured:
    slower than 16 threads).  So: the cores this job may use (scheduler affinity), capped at that share of 16; HICDIFF_CPU_THREADS overrides
    the cap.  The JSON line carries both figures (`cores` = threads used, `host_cpu_count` = os.cpu_count())."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(cores, int(os.environ.get("HICDIFF_CPU_THREADS", "16"))))


def build_model(w, device, beta_schedule="linear"):
    torch.manual_seed(1234)
    if w["arch"] == "unet":
        if w["cond"]:
            from hicdiff_amd.hicdiff_condition import GaussianDiffusion, Unet
            net = Unet(64, dim_mults=(1, 2, 4, 8), self_condition=True)
        else:
            from hicdiff_amd.hicdiff import GaussianDiffusion, Unet
            net = Unet(64, dim_mults=(1, 2, 4, 8))
    else:
        from hicdiff_amd.model.hicedrn_Diff import hicedrn_Diff
        if w["cond"]:
            from hicdiff_amd.hicdiff_condition import GaussianDiffusion
        else:
            from hicdiff_amd.hicdiff import GaussianDiffusion
        net = hicedrn_Diff(self_condition=w["cond"])
    net = net.to(device)
    diff = GaussianDiffusion(net, image_size=w["S"], timesteps=T_CHAIN, loss_type="l2", beta_schedule=beta_schedule).to(device)
    return net, diff


def cpu_baseline(w, budget_s=20.0):
    """The oracle (CPU port of the reference path) timed on this box's host cores on a bounded sample:
    the same step (eps-net + posterior update) on a small batch of the same tile size."""
    from oracle import diffusion as OD, nets as ON, weights as W
    cores = host_threads()
    torch.set_num_threads(cores)
    if w["arch"] == "unet":
        cfg = ON.UnetCfg(self_condition=w["cond"])
        sd = W.fill_state_dict(W.unet_shapes(self_condition=w["cond"]))
        bs = 8
    else:
        cfg = ON.HicedrnCfg(self_condition=w["cond"])
        sd = W.fill_state_dict(W.hicedrn_shapes(self_condition=w["cond"]))
        bs = 1
    model = ON.make_eps_fn(sd, cfg)
    kind = "cond" if w["cond"] else "uncond"
    ref = OD.DiffusionRef(model, image_size=w["S"], timesteps=T_CHAIN, beta_schedule="linear", loss_type="l2", kind=kind)
    g = torch.Generator().manual_seed(0)
    x = torch.randn((bs, 1, w["S"], w["S"]), generator=g)
    cond = torch.rand((bs, 1, w["S"], w["S"]), generator=g) * 2 - 1 if w["cond"] else None
    z = torch.randn(x.shape, generator=g)
    x, _, _ = ref.p_sample(x, T_CHAIN - 1, cond, z)        # warm-up step
    t0 = time.perf_counter()
    n = 0
    while True:
        x, _, _ = ref.p_sample(x, T_CHAIN - 2 - n, cond, z)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 50:
            break
    sec_per_step = el / n
    return {
        "value": bs / (T_CHAIN * sec_per_step), "unit": "tiles/s", "cores": cores, "host_cpu_count": os.cpu_count(), "kind": "port",
        "sample": f"{n} reverse steps (eps-net + posterior update) of the same {w['arch']} on {bs} tiles of "
                  f"1x{w['S']}x{w['S']}, torch CPU fp32 oracle, {cores} threads; scaled to a {T_CHAIN}-step chain",
    }


def bench_train(args):
    """--workload hicedrn64_train: training tiles / second of the native hicedrn step.
    A step = `loss = diffusion([lq, hq]); loss.backward(); optimizer.step(); optimizer.zero_grad()` (train.py:120-134) on synthetic
    tiles resident in HBM; data parallel under torchrun (the flat gradient is summed over the ranks stage by stage while the backward pass
    still runs: hicdiff_amd/_training.py StagedReducer), weak scaling."""
    rank, dev, world, dist = rank_env(args)
    from hicdiff_amd.hicdiff_condition import GaussianDiffusion
    from hicdiff_amd.model.hicedrn_Diff import hicedrn_Diff
    from hicdiff_amd.optim import Adam
    batch, tile, blocks = args.batch or 64, args.tile or 64, args.blocks
    torch.manual_seed(1234)
    if args.train_arch == "unet":
        from hicdiff_amd.hicdiff_condition import Unet
        net = Unet(64, dim_mults=(1, 2, 4, 8), self_condition=True)
    else:
        net = hicedrn_Diff(number_resnet=blocks, self_condition=True)
    d = GaussianDiffusion(net, image_size=tile, timesteps=1000, loss_type="l2", beta_schedule="linear").to(dev)
    d.model.train_precision = args.train_precision
    d.train()
    opt = Adam(d.parameters(), lr=2e-5)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    hq = torch.rand((batch, 1, tile, tile), device=dev, generator=g) * 2 - 1
    lq = (hq + 0.1 * torch.randn(hq.shape, device=dev, generator=g)).clamp(-1, 1)

    def step():
        loss = d([lq, hq])
        loss.backward()
        opt.step()
        opt.zero_grad()
        return loss

    first = None
    for _ in range(max(args.warmup, 1)):
        v = float(step().detach())
        first = v if first is None else first
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = max_over_ranks(time.perf_counter() - t0, dist, dev)
    # per-kernel durations: one more step with a HIP event pair around every convolution / GEMM / rewrite launch
    from hicdiff_amd import _lib as L
    lib = L.load()
    lib.hd_profile_enable(1)
    step()
    torch.cuda.synchronize()
    rows_buf = (L.HdProfileRow * L.HD_PROFILE_MAX_ROWS)()
    n_rows = lib.hd_profile_read(rows_buf, L.HD_PROFILE_MAX_ROWS)
    prof = {rows_buf[i].kernel.decode(): rows_buf[i] for i in range(max(n_rows, 0))}
    lib.hd_profile_enable(0)
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import diffusion as OD, nets as ON, train as OTR
        n = 2
        torch.set_num_threads(host_threads())
        sd = {k: v.detach().cpu().clone() for k, v in d.model.state_dict().items()}
        cfg = ON.UnetCfg(self_condition=True) if args.train_arch == "unet" else ON.HicedrnCfg(number_resnet=blocks, self_condition=True, sr3=False)
        buf = OD.diffusion_buffers("linear", 1000)
        m, v = {k: torch.zeros_like(p) for k, p in sd.items()}, {k: torch.zeros_like(p) for k, p in sd.items()}
        c0 = time.perf_counter()
        _, gr = OTR.loss_and_grads(sd, cfg, buf, hq[:n].cpu(), torch.randint(0, 1000, (n,)), torch.randn((n, 1, tile, tile)), lq[:n].cpu(), "l2")
        OTR.adam_step(sd, gr, m, v, 1)
        cpu = {"value": round(n / (time.perf_counter() - c0), 3), "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"one training step (autograd forward + backward + Adam) of the same net on {n} tiles, torch CPU fp32 oracle"}
    if rank == 0:
        ms = dt / args.steps * 1e3
        flop_tile = 2 * 9 * 256 * 256 * tile * tile * (2 * blocks + 1) * 3        # forward + data gradient + weight gradient of the 256->256 convs
        if args.train_arch == "unet":
            flop_tile = 3 * 14.384e9 * (tile / 64.0) ** 2                        # 3 x the forward's algorithmic flops
        achieved = flop_tile * batch / (ms / 1e3) / 1e12
        kernels = {}
        for name, r in prof.items():
            if r.launches and r.total_ms > 0:
                kernels[name] = {"launches": int(r.launches), "avg_launch_us": round(r.total_ms / r.launches * 1e3, 1),
                                 "TFLOPs": round(r.flops / r.total_ms / 1e9, 1), "GBps": round(r.bytes / r.total_ms / 1e6)}
        dom = max(kernels, key=lambda k: kernels[k]["launches"] * kernels[k]["avg_launch_us"]) if kernels else None
        per_product = 3 if args.train_precision == "bf16x3" else 1
        print(json.dumps({
            "metric": f"training tiles/sec ({args.train_arch}, l2, Adam)", "value": round(batch * world / (ms / 1e3), 2), "unit": "tiles/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 master weights and gradients; products " + ("split-bf16 x3 MFMA" if args.train_precision == "bf16x3" else "bf16 MFMA (one per product)") + ", fp32 accumulate",
            "data": "synthetic",
            "config": {"workload": (f"hicedrn64_train: hicedrn x{blocks} blocks" if args.train_arch == "hicedrn" else "hicedrn64_train --train-arch unet: UNet(64, (1,2,4,8))") +
                                   f", conditional, {batch} tiles of 1x{tile}x{tile} per GPU, Adam lr 2e-5",
                       "tiles_per_gpu": batch, "tile": tile, "parallelism": f"data-parallel x{world}, gradients summed stage by stage behind the backward pass (StagedReducer)"},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": kernels[dom]["TFLOPs"] if dom else None, "peak": round(PEAK_BF16_MFMA_TFLOPS / per_product, 1),
                         "unit": "TFLOP/s", "frac": round(kernels[dom]["TFLOPs"] / (PEAK_BF16_MFMA_TFLOPS / per_product), 4) if dom else None, "traffic": None,
                         "avg_launch_us": kernels[dom]["avg_launch_us"] if dom else None,
                         "peak_note": f"algorithmic TFLOP/s; peak = dense bf16 MFMA 2500 / {per_product} MFMA(s) per product",
                         "whole_step": {"achieved": round(achieved, 1), "frac": round(achieved / (PEAK_BF16_MFMA_TFLOPS / per_product), 4),
                                        "note": "forward + data gradient + weight gradient of the 256->256 convolutions over the step time"},
                         "kernels": kernels},
            "cpu_baseline": cpu, "loss_first_last": [first, float(last.detach())]}))
    if dist is not None:
        dist.destroy_process_group()


def bench_tiles(args):
    """--workload tiles: the tile producer / stitcher on one chromosome-sized matrix (chr1 at 10 kb).
    Algorithmic bytes: split reads and writes every tile element once (2 * ntiles * piece^2 * 4 B); stitch reads every tile once and
    writes the whole n x n matrix.  Kernels timed with HIP events on the launch stream, operands and tables resident in HBM."""
    import ctypes as C

    import numpy as np
    from hicdiff_amd import _lib as L
    from hicdiff_amd import processdata as PD
    from hicdiff_amd.processdata.PrepareData_linear_sing import stitch_table
    n, piece, res, reps = args.matrix_size, args.tile or 64, args.res, max(args.steps, 1)
    dev = torch.device("cuda:0")
    m = torch.rand((n, n), device=dev)
    m = (m + m.T) / 2
    tiles, org = PD.split_pieces_device(m, piece, piece, res)
    back = PD.stitch_pieces_device(tiles, org, n)
    torch.cuda.synchronize()

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    lib = L.load()
    o_dev = torch.from_numpy(org.astype(np.int32)).to(dev)
    tb = stitch_table(org, n, piece)
    t_dev = torch.from_numpy(tb).to(dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    P = lambda x: C.c_void_p(x.data_ptr())
    ms_split = timed(lambda: lib.hd_split_pieces(P(m), n, P(o_dev), len(org), piece, P(tiles), st))
    ms_stitch = timed(lambda: lib.hd_stitch_pieces(P(tiles), P(t_dev), tb.shape[0], piece, piece, P(back), n, st))
    ms_api = timed(lambda: PD.split_pieces_device(m, piece, piece, res))
    nt, pp = len(org), piece * piece
    b_split, b_stitch = 2 * nt * pp * 4, nt * pp * 4 + n * n * 4
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import tiles as OT
        ncpu = min(n, 6000)
        mc = m[:ncpu, :ncpu].cpu().numpy()
        t0 = time.perf_counter()
        ref = OT.split_pieces(mc, piece, piece, res)
        cpu = {"value": round(len(ref) / (time.perf_counter() - t0)), "unit": "tiles/s", "cores": 1, "kind": "port",
               "sample": f"numpy slicing (as the reference does) of the leading {ncpu}x{ncpu} block: {len(ref)} tiles"}
    print(json.dumps({
        "metric": "tiles cut / second (splitPieces)", "value": round(nt / ms_split * 1e3), "unit": "tiles/s", "n_gpus": 1, "steps": reps, "warmup": 1,
        "ms_per_step": round(ms_split, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"tiles: n={n} piece={piece} res={res}: {nt} tiles"},
        "roofline": {"bound": "hbm", "achieved": round(b_split / ms_split / 1e6, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": round(b_split / ms_split / 1e6 / PEAK_HBM_GBS, 3), "traffic": None, "kernel": "split_pieces_kernel<true>"},
        "stitch": {"ms": round(ms_stitch, 4), "GBps": round(b_stitch / ms_stitch / 1e6, 1), "frac_hbm": round(b_stitch / ms_stitch / 1e6 / PEAK_HBM_GBS, 3),
                   "kernel": "stitch_pieces_vec4_kernel"},
        "split_ms_with_host_tables": round(ms_api, 4), "cpu_baseline": cpu,
        "round_trip_exact": bool(torch.equal(PD.split_pieces_device(back, piece, piece, res)[0], tiles))}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="unet64", choices=sorted(WORK) + ["hicedrn64_train", "tiles"])
    ap.add_argument("--tile", type=int, default=None, help="hicedrn64_train / tiles: tile size (default 64)")
    ap.add_argument("--blocks", type=int, default=32, help="hicedrn64_train: residual blocks")
    ap.add_argument("--train-arch", choices=["hicedrn", "unet"], default="hicedrn", help="hicedrn64_train: which eps-network to train")
    ap.add_argument("--train-precision", choices=["bf16x3", "bf16"], default="bf16x3", help="hicedrn64_train: products of the convolutions (bf16: one MFMA per product)")
    ap.add_argument("--matrix-size", type=int, default=24896, help="tiles: matrix side (chr1 at 10 kb)")
    ap.add_argument("--res", type=int, default=10000, help="tiles: bin size")
    ap.add_argument("--batch", type=int, default=None, help="tiles per GPU (default: the workload's)")
    ap.add_argument("--total-tiles", type=int, default=None,
                    help="strong-scaling form (BASELINE configs[3]: 256 hicedrn tiles sharded over the node): the job is this many tiles in all, "
                         "total/N per GPU; reported with \"scaling\": \"strong\"")
    ap.add_argument("--chains", type=int, default=None, choices=[1, 2, 3, 4],
                    help="force one whole-batch chain or two half-batch chains per GPU (default: the library's rule, two for every replayed step: from 150 k pixels per step on)")
    ap.add_argument("--beta-schedule", default="linear", choices=["linear", "cosine", "sigmoid"],
                    help="beta schedule of the chain (default: train.py's linear, the headline configuration since round 1; the precision schedule "
                         "applies to it alone -- sigmoid / cosine chains run split-bf16 x3 at every step, DESIGN.md section 4e)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-table", action="store_true", help="also print one line per convolution kernel (stderr): launches, ms per step, TFLOP/s-eq, GB/s")
    ap.add_argument("--full-chain", action="store_true", help="time a whole chain: --steps becomes 1000 (t = 999 .. 0)")
    ap.add_argument("--sustained-budget", type=float, default=45.0,
                    help="seconds the post-timing sustained run may take (a whole 1000-step chain when it fits; 0 disables)")
    args = ap.parse_args()
    if args.full_chain:
        args.steps = T_CHAIN
    if args.total_tiles and (args.batch or args.total_tiles % args.gpus):
        raise SystemExit("--total-tiles: the tiles are dealt evenly, total % gpus must be 0, and --batch is then implied")
    if args.gpus > 1 and "RANK" not in os.environ:
        # not under a launcher: start the N rank processes here.  This process has not touched the GPU and never will.
        from hicdiff_amd.sharding import launch_ranks
        sys.exit(launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    if args.workload == "hicedrn64_train":
        return bench_train(args)
    if args.workload == "tiles":
        return bench_tiles(args)

    w = dict(WORK[args.workload])
    if args.batch:
        w["B"] = args.batch
    if args.total_tiles:
        w["B"] = args.total_tiles // args.gpus
    rank, device, world, dist = rank_env(args)

    from hicdiff_amd import _lib as L
    from hicdiff_amd.sharding import all_gather_tiles
    lib = L.load()
    net, diff = build_model(w, device, args.beta_schedule)
    B, S = w["B"], w["S"]
    diff.tile_offset = rank * B                       # noise keyed by the GLOBAL tile index
    img = diff._initial_noise((B, 1, S, S), device)
    cond = (torch.rand((B, 1, S, S), device=device) * 2 - 1) if w["cond"] else None

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    eng = net.engine(device)
    if args.chains is not None:
        eng.set_chains(args.chains)
    chains = eng.chains_for(B, S)

    def run_ts(ts):
        """One reverse step at each timestep of `ts`, inside the sampler's chain bracket (hicdiff_amd/_diffusion.py:_ancestral): the state
        meets this stream at the ends only.  (A step's cost does not depend on the state it is given.)"""
        with eng.chain(B, S):
            for t in ts:
                diff._step_inplace(img, t, cond, eng=eng)

    def consecutive(n, t):
        """n consecutive timesteps from t downwards (wrapping into a new chain below 0)."""
        return [(t - i) % T_CHAIN for i in range(n)]

    def spread(n):
        """n timesteps spread evenly over the chain, t = 999 first: since round 4 a step's cost depends on the HALF of the chain it sits in
        (the precision schedule), so K consecutive steps from t = 999 would time the cheaper half only.  The K-step
        region is a stratified sample of the chain -- exactly the chain's mix of the three arithmetics when K is a multiple of 4 (the default 20 is) -- and
        `sustained` below is a whole chain."""
        return [T_CHAIN - 1 - (i * T_CHAIN) // n for i in range(n)]

    # set-up, not warm-up: the step's hipGraphs are captured on the second call of each arithmetic (engine.hip lane_step)
    for t in (T_CHAIN - 1, (T_CHAIN * 5) // 8, 0):
        run_ts([t, t, t])
    timed = consecutive(T_CHAIN, T_CHAIN - 1) if args.steps >= T_CHAIN else spread(args.steps)
    run_ts(spread(args.warmup))
    barrier()
    t0 = time.perf_counter()
    run_ts(timed)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, dist, device)
    sec_per_step = elapsed / args.steps
    value = (B * world) / (T_CHAIN * sec_per_step)
    early = sum(1 for t in timed if diff._early_band(t))

    # Sustained rate: a whole chain (all ranks, same barriers), with HIP events on the launch stream after 20 steps
    # and at the end, so the first-20 burst and the steady state of the SAME run can be compared (clock give-back).
    sustained = None
    n_sus = int(min(T_CHAIN, args.sustained_budget / max(sec_per_step, 1e-9)))
    if args.steps >= T_CHAIN:
        n_sus = 0                                     # the timed region already was a whole chain
    if n_sus >= 100:
        burst = min(20, n_sus)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        barrier()
        s0 = time.perf_counter()
        ev[0].record()
        run_ts(consecutive(burst, T_CHAIN - 1))
        ev[1].record()
        run_ts(consecutive(n_sus - burst, T_CHAIN - 1 - burst))
        ev[2].record()
        barrier()
        sus_s = max_over_ranks(time.perf_counter() - s0, dist, device)
        sustained = {"steps": n_sus, "ms_per_step": round(sus_s / n_sus * 1e3, 4),
                     "tiles_per_s": round((B * world) / (T_CHAIN * sus_s / n_sus), 4),
                     "first_20_ms_per_step": round(ev[0].elapsed_time(ev[1]) / burst, 4),
                     "rest_ms_per_step": round(ev[1].elapsed_time(ev[2]) / max(n_sus - burst, 1), 4),
                     "note": "one chain from t=999 timed after the K-step region (rank 0 events; ms_per_step is max over ranks, host clock); "
                             "first_20 lies in the early band of the precision schedule, so it is cheaper than the chain's mean"}

    # Roofline of the dominant kernel: the same step, launched eagerly with a HIP event pair around every
    # convolution launch on its stream (the timed region above replays a captured hipGraph of the step,
    # inside which per-launch events cannot be recorded).
    prof_steps = max(2, min(args.steps, 6) // 2 * 2)
    lib.hd_profile_enable(1)
    run_ts(spread(prof_steps))
    barrier()
    rows_buf = (L.HdProfileRow * L.HD_PROFILE_MAX_ROWS)()
    n_rows = lib.hd_profile_read(rows_buf, L.HD_PROFILE_MAX_ROWS)
    rows = [rows_buf[i] for i in range(max(n_rows, 0))]
    lib.hd_profile_enable(0)
    if args.kernel_table and rank == 0:
        for r in sorted(rows, key=lambda r: -r.total_ms):
            ms = r.total_ms / prof_steps
            print(f"{r.kernel.decode():58s} {r.launches // prof_steps:4d}/step {ms:7.3f} ms/step {r.flops / r.total_ms / 1e9:7.1f} TFLOP/s-eq "
                  f"{r.bytes / r.total_ms / 1e6:7.0f} GB/s", file=sys.stderr)

    # the one collective of the path: gather every rank's finished tiles (rank-ordered)
    gather_ms = None
    if dist is not None:
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        src = img if dist.get_backend() == "nccl" else img.cpu()
        full = all_gather_tiles(src, dist)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        assert full.shape[0] == B * world

    if rank == 0:
        # A kernel's roofline follows its arithmetic: three bf16 MFMAs per product (2500 / 3), two fp16 MFMAs (2500 / 2: the early band's 3x3
        # kernels), or the exact-fp32 MFMA.
        def peak_of(name):
            return (PEAK_BF16_MFMA_TFLOPS if b"f16w1" in name else PEAK_BF16_MFMA_TFLOPS / 2.0 if b"f16w2" in name else
                    PEAK_BF16_MFMA_TFLOPS / 3.0 if b"bf16x3" in name else PEAK_F32_MFMA_TFLOPS)

        dom = max(rows, key=lambda r: r.total_ms)
        ach = dom.flops / (dom.total_ms * 1e-3) / 1e12 if dom.total_ms > 0 else 0.0
        split = b"bf16x3" in dom.kernel or b"f16w" in dom.kernel
        peak = peak_of(dom.kernel)
        traffic, traffic_src = recorded_traffic(args.workload, dom.kernel.decode(), B, WORK[args.workload]["B"])
        whole_flops = w["flop"] * B / sec_per_step / 1e12
        # the step's own MFMA roofline: the chain's mix of the two arithmetics over the timed steps (1x1 convolutions always take three products;
        # priced with the 3x3 layers, i.e. slightly in the step's favour -- they are 5 % of the flops)
        # below the band the low-resolution 3x3 layers (42 % of the matrix work: tests/studies/error_budget_study.py) take two products when the
        # schedule says so: 1 / (0.42 / (2500/2) + 0.58 / (2500/3)) = 969 TFLOP/s-eq for such a step, i.e. 2.58 products per multiply
        late = 2.58 if (early and diff.late_band_low_f16 and w["arch"] == "unet") else 3.0
        per_product = [1.0 if (diff._early_band(t) and t >= int(diff.early_band_x1_from * T_CHAIN)) else 2.0 if diff._early_band(t) else late for t in timed]
        step_peak = sum(PEAK_BF16_MFMA_TFLOPS / p for p in per_product) / len(timed) if split else PEAK_F32_MFMA_TFLOPS
        conv_ms = sum(r.total_ms for r in rows)
        roofline = {
            "bound": "mfma", "achieved": round(ach, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(ach / peak, 4), "traffic": traffic, "traffic_unit": "bytes per launch (2*FETCH_SIZE + WRITE_SIZE)",
            "traffic_source": traffic_src, "algorithmic_bytes_per_launch": round(dom.bytes / max(dom.launches, 1)),
            "peak_note": ("algorithmic fp32-equivalent TFLOP/s; peak = dense bf16 / fp16 MFMA 2500 / MFMAs per product (3: split-bf16 x3; 2 / 1: the early "
                          "band's fp16 xh (wh + wl) / xh wh)" if split else "exact-fp32 MFMA peak"),
            "kernel": dom.kernel.decode(), "launches": int(dom.launches),
            "launch_note": ("per-launch figures are HIP events around WHOLE-BATCH eager launches of the profiled steps (the kernel on its own, as with --chains 1: "
                            "the rocprofv3 summary to compare with is profiles/*_chains1_kernel_stats.csv); the timed region replays the same kernels on "
                            f"{chains} sub-batch chains whose launches overlap in a trace" if chains > 1 else "per-launch HIP events of the profiled eager steps"),
            "avg_launch_us": round(dom.total_ms * 1e3 / max(dom.launches, 1), 2),
            "conv_time_share": round(conv_ms * 1e-3 / (prof_steps * sec_per_step), 4),
            "all_convs_frac": round(sum(r.flops / peak_of(r.kernel) for r in rows) / max(conv_ms * 1e-3, 1e-12) / 1e12, 4),
            "profiled_steps": prof_steps,
            "whole_step": {
                "algorithmic_TFLOPs": round(whole_flops, 1),
                "frac_of_mfma_roofline": round(whole_flops / step_peak, 4),
                "mfma_roofline_TFLOPs": round(step_peak, 1),
                "hbm_frac_algorithmic": round((w["act_bytes"] * B + w["w_bytes"]) / sec_per_step / 1e9 / PEAK_HBM_GBS, 4),
                "note": "SURVEY 8(d) per-tile-step flops / bytes x tiles over the measured step time; MFMA roofline = the timed steps' mix of 2500 / products "
                        "per multiply (1 and 2 in the early band, 2.58 or 3 below it), HBM = 8 TB/s (BASELINE target: hbm_frac_algorithmic >= 0.5)",
            },
        }
        out = {
            "metric": "denoised Hi-C tiles/sec (1000-step reverse)", "value": round(value, 4), "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(sec_per_step * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if args.total_tiles else "weak", "vs_baseline": None,
            "dtype": ("f32 (wide convs: split-bf16 x3 MFMA, fp32 accumulate" + ("; 3x3 convs of the steps t >= T/2: two fp16 MFMAs per product, t >= 3T/4: one" if early else "") + ")")
                     if split else "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {w['arch']} eps-net, {'conditional' if w['cond'] else 'unconditional'}, "
                                   f"1x{S}x{S} tiles, {B} tiles/GPU, ancestral DDPM T={T_CHAIN}, {args.beta_schedule} beta schedule, device Philox noise",
                       "tiles_per_gpu": B, "tile": S, "chain_steps": T_CHAIN, "beta_schedule": args.beta_schedule, "parallelism": f"tile-shard x{world}",
                       "chains": chains,
                       **({"total_tiles": args.total_tiles} if args.total_tiles else {}),
                       "precision_schedule": (f"3x3 convs: one fp16 product per multiply for t >= {int(diff.early_band_x1_from * T_CHAIN)}, two for t >= "
                                              f"{int(diff._band_from() * T_CHAIN)}, split-bf16 x3 below" +
                                              (" except two products on the maps of at most (S/4)^2 pixels" if late != 3.0 else "") +
                                              f" ({early} of the {len(timed)} timed steps in the early band, "
                                              f"{sum(1 for p in per_product if p == 1.0)} of them on one product)" if early else "split-bf16 x3 at every step"),
                       "timed_region": ("one whole chain, t = 999 .. 0" if args.steps >= T_CHAIN else
                                        f"{args.steps} steps spread evenly over the chain, t = 999, {timed[1] if len(timed) > 1 else 999}, ... (a stratified sample: "
                                        "the two halves of the chain cost differently; see `sustained` for a whole chain)")},
            "roofline": roofline,
        }
        if sustained is not None:
            out["sustained"] = sustained
        if gather_ms is not None:
            out["all_gather_ms"] = round(gather_ms, 3)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
