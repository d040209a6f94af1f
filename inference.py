#!/usr/bin/env python3
"""Inference driver: same command line as the reference's ``inference.py`` (flags :22-36), running the
MI355X engine.

    python inference.py -u 1 -b 64 -l Human -n 1 -s 0.1           # DDRM 'deno' path (metrics_diff.py:121-224)
    python inference.py -u '' -b 64                                # conditional path (metrics_cond.py:61-137)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 inference.py -u 1 ...

What differs from the reference, deliberately (SURVEY.md section 3.1 lists the upstream quirks):
  * ``-u`` keeps the reference's ``type=bool`` behaviour (any non-empty string, including "0", selects the
    unsupervised branch; only ``-u ''`` selects the conditional one); ``--conditional`` says it explicitly;
  * the unsupervised branch upstream passes timestep=2000 to a 1000-step table and cannot run; here the DDRM
    stride comes from ``--sampling-steps`` (default 1000 -> every step, 50 -> stride 20 = BASELINE config 1);
  * the data pipeline (cooler/.mcool, processdata/) is out of scope: tiles are synthetic Hi-C-like
    matrices (``--synthetic N``) or ``.npy`` arrays given with ``--noisy`` / ``--target``;
  * checkpoints are optional (``--weights``); without one the net has seeded random weights;
  * under torch.distributed the tile list is sharded over ranks and gathered by one RCCL all-gather.
Outputs: ``Outputs_diff/<name>/{predict,target,noisy,inds}.npy`` as upstream (metrics_diff.py:203-210).
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def create_parser():
    p = argparse.ArgumentParser(description="HiCDiff inference on MI355X")
    # the reference's six flags, verbatim (inference.py:22-36)
    p.add_argument("-u", "--unspervised", type=bool, default=True, help="True: unsupervised DDRM denoising; '' : conditional")
    p.add_argument("-b", "--batch_size", type=int, default=64)
    p.add_argument("-e", "--epoch", type=int, default=400)
    p.add_argument("-l", "--celline", type=str, default="Human")
    p.add_argument("-n", "--celln", type=int, default=1)
    p.add_argument("-s", "--sigma", type=float, default=1)
    # explicit additions
    p.add_argument("--conditional", action="store_true", help="conditional (low-coverage -> high-coverage) sampling")
    p.add_argument("--arch", choices=["hicedrn", "unet"], default="hicedrn")
    p.add_argument("--resnet-blocks", type=int, default=32)
    p.add_argument("--tile", type=int, default=64)
    p.add_argument("--timesteps", type=int, default=1000)
    p.add_argument("--sampling-steps", type=int, default=1000, help="DDRM steps (stride = 1000 // steps)")
    p.add_argument("--schedule", default="sigmoid", choices=["linear", "sigmoid", "cosine"])
    p.add_argument("--synthetic", type=int, default=64, help="number of synthetic tiles when no --noisy is given")
    p.add_argument("--noisy", default=None, help=".npy of low-coverage tiles (N,1,S,S) in [-1,1]")
    p.add_argument("--target", default=None, help=".npy of high-coverage tiles (N,1,S,S)")
    p.add_argument("--matrix", default=None, help="Full_Mats .npy of one chromosome in [-1,1]: cut into --tile tiles on the GPU (splitPieces), "
                   "degraded with --sigma as split_numpy does, denoised, and stitched back into <out>/predict_matrix.npy")
    p.add_argument("--data-root", default=None, help="directory holding DataFull/DataFull_<cell>_cell<n>_40000_deno_<sigma>/Splits: the reference's own "
                   "flow (inference.py:104-118) -- VisionMetrics.getMetrics over the test chromosomes, inds.npy = chromosome of every tile")
    p.add_argument("--chro", default="test", help="--data-root: 'test' or one chromosome number")
    p.add_argument("--res", type=int, default=40000, help="bin size of --matrix (sets the band of tiles, PrepareData_linear_sing.py:31,42)")
    p.add_argument("--weights", default=None, help="state_dict written by the reference's train.py")
    p.add_argument("--outdir", default=os.path.join(ROOT, "Outputs_diff"))
    p.add_argument("--seed", type=int, default=1234)
    p.add_argument("--precision", choices=["bf16x3", "f32"], default="bf16x3")
    p.add_argument("--skip-inert-steps", action="store_true",
                   help="DDRM denoising: do not evaluate the network on the steps whose update ignores it (etaB = 1 while sigma_next > sigma_0: with the "
                        "defaults 47 of 50 steps); the tiles are bit-identical, only faster.  Off by default: the reference evaluates every step")
    p.add_argument("--metrics", action="store_true", help="report mse / psnr / ssim / snr / pcc of predict vs target (GPU, stard_metrics.py:146-160)")
    return p


def synthetic_tiles(n, s, sigma_0, seed):
    """Hi-C-like tiles: hq = sym(2*U^3 - 1); noisy = clip(hq + sigma_0 * N(0,1))."""
    g = torch.Generator().manual_seed(seed)
    a = 2 * torch.rand((n, 1, s, s), generator=g) ** 3 - 1
    hq = (a + a.transpose(-1, -2)) / 2
    lq = (hq + sigma_0 * torch.randn((n, 1, s, s), generator=g)).clamp(-1, 1)
    return lq, hq


def build_diffusion(args, conditional, S, device, rank=0):
    torch.manual_seed(args.seed)
    if args.arch == "hicedrn":
        from hicdiff_amd.model.hicedrn_Diff import hicedrn_Diff
        net = hicedrn_Diff(number_resnet=args.resnet_blocks, self_condition=conditional)
    else:
        from hicdiff_amd.hicdiff import Unet
        net = Unet(64, dim_mults=(1, 2, 4, 8), self_condition=conditional)
    if conditional:
        from hicdiff_amd.hicdiff_condition import GaussianDiffusion
    else:
        from hicdiff_amd.hicdiff import GaussianDiffusion
    diffusion = GaussianDiffusion(net, image_size=S, timesteps=args.timesteps, loss_type="l2", beta_schedule=args.schedule)
    if args.weights:
        diffusion.load_state_dict(torch.load(args.weights, map_location="cpu"))
    elif rank == 0:
        print("[inference] no --weights given: seeded random weights (throughput / plumbing runs only)")
    diffusion = diffusion.to(device).eval()
    diffusion.seed = args.seed
    return diffusion


def run_test_split(args, conditional, device):
    """The reference's Inference() (inference.py:38-118): build the model, hand it to VisionMetrics.getMetrics, which walks the test
    split through the DataModule and writes Outputs_diff/<name>/{predict,target,noisy,inds}.npy."""
    diffusion = build_diffusion(args, conditional, args.tile, device)
    chro = int(args.chro) if str(args.chro).isdigit() else args.chro
    name = ("hicedrn_l2_" if args.arch == "hicedrn" else "unet_l2_") + args.schedule[:3]
    out_root = os.path.dirname(os.path.abspath(args.outdir)) if os.path.basename(os.path.normpath(args.outdir)) == "Outputs_diff" else args.outdir
    if conditional:
        from hicdiff_amd.Utils import metrics_cond as vm_cond
        vmx = vm_cond.VisionMetrics(image_channel=1, image_size=args.tile, timestep=args.timesteps, type="condition")
        predict = vmx.getMetrics(model=diffusion.super_resolution, model_name=name, device=device, chro=chro, deg="deno", sigma=args.sigma,
                                 cellN=args.celln, cell_line=args.celline, root=args.data_root, outdir=out_root)
    else:
        from hicdiff_amd.Utils import metrics_diff as vm
        vmx = vm.VisionMetrics(image_channel=1, image_size=args.tile, sehedule=args.schedule, timestep=args.sampling_steps)
        vmx.seed = args.seed
        predict = vmx.getMetrics(model=diffusion.model, model_name=name, device=device, chro=chro, deg="deno", sigma=args.sigma,
                                 cellN=args.celln, cell_line=args.celline, root=args.data_root, outdir=out_root)
    r = vmx.last_result
    print(f"[inference] {r['nsamples']} tiles -> {vmx.last_dir}")
    if args.metrics and r["nsamples"]:
        print(f"[inference] predict vs target: mse {r['mse'] / r['nsamples']:.5f}, psnr {r['psnr']:.4f}, ssim {r['ssim']:.4f}, pcc {r['pcc']:.4f}")
    return torch.from_numpy(predict)


def _store_barrier(dist, rank, world, key="hicdiff_eval_done", poll_s=5.0):
    """Rank 0 posts a key on the rendezvous store when it is done, the others poll for it: no collective is in flight while they wait, so no
    collective watchdog can fire however long rank 0 takes."""
    import time
    store = dist.distributed_c10d._get_default_store()
    if rank == 0:
        store.set(key, "1")
        return
    while True:
        try:
            if store.check([key]):
                return
        except RuntimeError:
            return                      # the store went away with rank 0: nothing left to wait for
        time.sleep(poll_s)


def main(argv=None):
    args = create_parser().parse_args(argv)
    conditional = args.conditional or not args.unspervised
    os.environ["HICDIFF_PRECISION"] = args.precision
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    if args.data_root:
        # The evaluation harness (VisionMetrics.getMetrics, upstream a single-process flow) prepares the Splits/ files and writes one set of
        # Outputs_diff/<name>/*.npy: under torchrun it runs on rank 0 only -- every rank running it would race on those files -- and the other
        # ranks wait at the common barrier.  (Sharded sampling of a tile list is the path below: --noisy / --matrix / --synthetic.)
        # The wait is a store barrier with a long timeout, reached in a `finally`: an NCCL barrier would abort the job after its watchdog's
        # ten minutes while rank 0 is still sampling a whole test split, and a rank 0 that raises must not leave the others hanging.
        predict = None
        try:
            predict = run_test_split(args, conditional, device) if rank == 0 else None
        finally:
            if dist is not None:
                import datetime
                dist.monitored_barrier(timeout=datetime.timedelta(hours=24)) if dist.get_backend() == "gloo" else _store_barrier(dist, rank, world)
                dist.destroy_process_group()
        return predict
    origins = None
    if args.matrix:
        from hicdiff_amd.processdata import split_pieces_device
        full = np.load(args.matrix)
        tiles_dev, origins = split_pieces_device(torch.from_numpy(np.ascontiguousarray(full, dtype=np.float32)).to(device), args.tile, args.tile, args.res)
        hq = tiles_dev.cpu()
        g = torch.Generator().manual_seed(args.seed)
        lq = hq + (args.sigma if args.sigma <= 1 else 0.1) * torch.randn(hq.shape, generator=g)     # :194-202 with deg='deno'
    elif args.noisy:
        lq = torch.from_numpy(np.load(args.noisy)).float()
        hq = torch.from_numpy(np.load(args.target)).float() if args.target else torch.zeros_like(lq)
    else:
        lq, hq = synthetic_tiles(args.synthetic, args.tile, args.sigma if args.sigma <= 1 else 0.1, args.seed)
    n, S = lq.shape[0], lq.shape[-1]

    diffusion = build_diffusion(args, conditional, S, device, rank)

    from hicdiff_amd.sharding import sample_sharded
    bs = args.batch_size

    def run_local(start, count):
        outs = []
        for b0 in range(start, start + count, bs):
            b1 = min(b0 + bs, start + count)
            y = lq[b0:b1].to(device)
            if conditional:                                   # metrics_cond.py:100-107
                diffusion.tile_offset = b0
                outs.append(diffusion.super_resolution(y))
            else:                                             # metrics_diff.py:165-224
                from hicdiff_amd.functions.denoising import efficient_generalized_steps
                from hicdiff_amd.functions.H_func import MakeFunc
                from hicdiff_amd._diffusion import _SCHEDULES
                betas = _SCHEDULES[args.schedule](1000).float().to(device)     # metrics_diff.py:36-81,100-109
                seq = range(0, 1000, max(1, 1000 // args.sampling_steps))
                x = diffusion.model.engine(device).randn(b1 - b0, S, args.seed, b0, 1 << 20)
                xs, _ = efficient_generalized_steps(x, seq, diffusion.model, betas, MakeFunc("deno", 1, S, device), y.reshape(b1 - b0, -1),
                                                    args.sigma if args.sigma <= 1 else 0.1, etaB=1.0, etaA=0.85, etaC=0.85,
                                                    keep="last", seed=args.seed, tile_offset=b0, skip_inert_steps=args.skip_inert_steps)
                outs.append(xs[-1])
        return torch.cat(outs) if outs else torch.empty((0, 1, S, S), device=device)

    predict = sample_sharded(run_local, n, dist)
    if rank == 0:
        sigma = args.sigma
        name = ("hicedrn_l2_" if args.arch == "hicedrn" else "unet_l2_") + args.schedule[:3] + args.celline + str(args.celln) + \
            "_deno_" + str(sigma) + "_trans2_" + str(args.sampling_steps if not conditional else args.timesteps)
        out = os.path.join(args.outdir, name)
        os.makedirs(out, exist_ok=True)
        np.save(os.path.join(out, "predict"), predict.cpu().numpy())
        np.save(os.path.join(out, "target"), hq.numpy())
        np.save(os.path.join(out, "noisy"), lq.numpy())
        np.save(os.path.join(out, "inds"), np.arange(n, dtype=np.int64))
        print(f"[inference] {n} tiles of 1x{S}x{S} -> {out}")
        if origins is not None:
            from hicdiff_amd.processdata import stitch_pieces_device
            np.save(os.path.join(out, "predict_matrix"), stitch_pieces_device(predict, origins, full.shape[0], args.tile).cpu().numpy())
            print(f"[inference] stitched {full.shape[0]}x{full.shape[0]} matrix -> {out}/predict_matrix.npy")
        if args.metrics:
            from hicdiff_amd.Utils.metrics import MetricLog
            log, base = MetricLog(), MetricLog()
            for b0 in range(0, n, bs):
                log.update(predict[b0:b0 + bs], hq[b0:b0 + bs].to(device))
                base.update(lq[b0:b0 + bs].to(device), hq[b0:b0 + bs].to(device))
            fmt = lambda r: f"mse {r['mse'] / r['nsamples']:.5f}, " + ", ".join(f"{k} {r[k]:.4f}" for k in ("psnr", "ssim", "pcc"))
            print(f"[inference] predict vs target: {fmt(log.r)}")
            print(f"[inference] noisy   vs target: {fmt(base.r)}")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return predict


if __name__ == "__main__":
    main()
