#!/usr/bin/env python3
"""Training-step throughput of the native hicedrn trainer on one GPU or under torchrun.

    python tools/bench_train.py [--batch 64] [--tile 64] [--blocks 32] [--steps 5] [--warmup 2] [--cond 1]

A step = `loss = diffusion([lq, hq]); loss.backward(); optimizer.step(); optimizer.zero_grad()` on synthetic tiles (the
reference's train.py:120-134).  Prints one JSON line: ms/step, tiles/s, algorithmic TFLOP/s (3 x the forward's 314 GFLOP per
tile: forward + data gradient + weight gradient), and the share of the weight-gradient GEMM when HICDIFF_PROFILE is set.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--blocks", type=int, default=32)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cond", type=int, default=1)
    ap.add_argument("--cpu-baseline", type=int, default=0, help="tiles for a bounded CPU leg (oracle autograd + Adam restatement, torch CPU fp32); 0 = skip")
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from hicdiff_amd.model.hicedrn_Diff import hicedrn_Diff
    from hicdiff_amd.optim import Adam
    if a.cond:
        from hicdiff_amd.hicdiff_condition import GaussianDiffusion
    else:
        from hicdiff_amd.hicdiff import GaussianDiffusion
    torch.manual_seed(1234)
    net = hicedrn_Diff(number_resnet=a.blocks, self_condition=bool(a.cond))
    d = GaussianDiffusion(net, image_size=a.tile, timesteps=1000, loss_type="l2", beta_schedule="linear").to(dev)
    d.train()
    opt = Adam(d.parameters(), lr=2e-5)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    hq = torch.rand((a.batch, 1, a.tile, a.tile), device=dev, generator=g) * 2 - 1
    lq = (hq + 0.1 * torch.randn(hq.shape, device=dev, generator=g)).clamp(-1, 1)

    def step():
        loss = d([lq, hq] if a.cond else hq)
        loss.backward()
        opt.step()
        opt.zero_grad()
        return loss

    losses = []
    for _ in range(a.warmup):
        losses.append(float(step().detach()))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        last = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    losses.append(float(last.detach()))
    cpu = None
    if rank == 0 and a.cpu_baseline > 0:
        from oracle import diffusion as OD, nets as ON, train as OTR
        n = a.cpu_baseline
        sd = {k: v.detach().cpu().clone() for k, v in d.model.state_dict().items()}
        cfg = ON.HicedrnCfg(number_resnet=a.blocks, self_condition=bool(a.cond), sr3=False)
        buf = OD.diffusion_buffers("linear", 1000)
        m, v = {k: torch.zeros_like(p) for k, p in sd.items()}, {k: torch.zeros_like(p) for k, p in sd.items()}
        tt = torch.randint(0, 1000, (n,))
        ee = torch.randn((n, 1, a.tile, a.tile))
        c0 = time.perf_counter()
        _, gr = OTR.loss_and_grads(sd, cfg, buf, hq[:n].cpu(), tt, ee, lq[:n].cpu() if a.cond else None, "l2")
        OTR.adam_step(sd, gr, m, v, 1)
        cdt = time.perf_counter() - c0
        cpu = {"value": round(n / cdt, 3), "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"one training step (autograd forward + backward + Adam) of the same net on {n} tiles, torch CPU fp32 oracle"}
    if rank == 0:
        ms = dt / a.steps * 1e3
        flop_tile = 2 * 9 * 256 * 256 * a.tile * a.tile * (2 * a.blocks + 1) * 3        # fwd + dgrad + wgrad of the 256->256 convs
        print(json.dumps({"metric": "training tiles/sec (hicedrn, l2, Adam)", "value": round(a.batch * world / (ms / 1e3), 2), "unit": "tiles/s",
                          "n_gpus": world, "ms_per_step": round(ms, 2), "steps": a.steps, "warmup": a.warmup,
                          "config": {"workload": f"hicedrn x{a.blocks} blocks, {'conditional' if a.cond else 'unconditional'}, {a.batch} tiles of 1x{a.tile}x{a.tile} per GPU"},
                          "algorithmic_TFLOPs": round(flop_tile * a.batch / (ms / 1e3) / 1e12, 1), "dtype": "f32 master, split-bf16 x3 MFMA products",
                          "loss_first_last": [losses[0], losses[-1]], "cpu_baseline": cpu}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
