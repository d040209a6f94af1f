#!/usr/bin/env python3
"""Training driver with the reference's command line (``train.py`` :28-42) on the MI355X engine.

The reference's loop (train.py:109-190): Adam(lr=2e-5) over ``diffusion.parameters()``; per batch
``loss = diffusion(x); loss.backward(); optimizer.step(); optimizer.zero_grad()``; per epoch a validation pass under
``no_grad``, ``bestg_*.pytorch`` on a new best validation loss, ``finalg_*.pytorch`` at the end, ``Epoch / train/loss /
valid/loss`` logged (to wandb upstream, as JSON lines here).

Both networks run the NATIVE step -- forward, backward and Adam are HIP kernels (include/hicdiff_hip.h "training step"): hicedrn (the
network upstream's train.py trains, conditional and unconditional) and the UNet (upstream: pretrain/train_unet_*.py).  Under torchrun every
rank takes its own batches and the flat gradient is all-reduced once per step (RCCL).  ``--eval-only`` evaluates the objective of the
frozen network instead.

Data: ``--data-root`` points at the directory that holds ``DataFull/`` (Splits written by hicdiff_amd.processdata or by the
reference); without it, synthetic Hi-C-like tiles.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def create_parser():
    p = argparse.ArgumentParser(description="HiCDiff training on MI355X")
    p.add_argument("-u", "--unspervised", type=bool, default=True)       # reference semantics: '' -> conditional
    p.add_argument("-b", "--batch_size", type=int, default=64)
    p.add_argument("-e", "--epoch", type=int, default=400)
    p.add_argument("-l", "--celline", type=str, default="Human")
    p.add_argument("-n", "--celln", type=int, default=1)
    p.add_argument("-s", "--sigma", type=float, default=1)               # train.py:38
    p.add_argument("--arch", choices=["hicedrn", "unet"], default="hicedrn")
    p.add_argument("--resnet-blocks", type=int, default=32)
    p.add_argument("--tile", type=int, default=64)
    p.add_argument("--tiles-per-epoch", type=int, default=256, help="synthetic tiles per epoch and split when no --data-root is given")
    p.add_argument("--data-root", default=None, help="directory holding DataFull/DataFull_<cell>_cell<n>_40000_deno_<sigma>/Splits")
    p.add_argument("--lr", type=float, default=2e-5)
    p.add_argument("--precision", choices=["bf16x3", "bf16"], default="bf16x3",
                   help="products of the convolutions: bf16x3 = fp32-equivalent (default, the arithmetic of the parity tests); bf16 = one bf16 MFMA per "
                        "product, fp32 accumulate and fp32 master weights (mixed-precision training, ~1.8x faster)")
    p.add_argument("--optimize", action="store_true", help="(kept for older command lines: the optimiser step is the default)")
    p.add_argument("--eval-only", action="store_true", help="loss curves of the frozen network, no optimiser step")
    p.add_argument("--weights-dir", default=os.path.join(ROOT, "Model_Weights"))
    p.add_argument("--seed", type=int, default=1234)
    p.add_argument("--print-checksum", action="store_true", help="every rank prints the sum and absolute sum of its parameters at the end")
    return p


class _Tiles:
    """The (low-coverage, target) tiles of both splits, loaded ONCE (the reference builds its DataLoaders once, train.py:75-84).
    Under torchrun rank 0 alone runs `prepare_data` (it may write the Splits/ files); the others wait at a barrier."""

    def __init__(self, args, rank, dist):
        self.args, self.sets = args, {}
        if not args.data_root:
            return
        from hicdiff_amd.processdata import GSE130711Module, GSE131811Module
        cls = GSE130711Module if args.celline == "Human" else GSE131811Module
        dm = cls(batch_size=args.batch_size, res=40000, piece_size=args.tile, cell_line=args.celline, cell_No=args.celln, sigma_0=args.sigma,
                 root=args.data_root)
        if rank == 0:
            dm.prepare_data()
        if dist is not None:
            dist.barrier()
        dm.setup("fit")
        self.sets = {"train": (dm.train_set.data, dm.train_set.target), "valid": (dm.val_set.data, dm.val_set.target)}

    def split(self, epoch, split):
        args = self.args
        if not self.sets:
            from inference import synthetic_tiles
            return synthetic_tiles(args.tiles_per_epoch, args.tile, args.sigma, args.seed + 2 * epoch + (split != "train"))
        lq, hq = self.sets[split]
        if split == "train":                                             # DataLoader(shuffle=True): a new order every epoch, the same on every rank
            order = torch.randperm(lq.shape[0], generator=torch.Generator().manual_seed(args.seed + epoch))
            return lq[order], hq[order]
        return lq, hq


def batch_plan(n, bs, rank, world, train):
    """Index lists of this rank's batches.  The reference's DataLoader has no drop_last (PrepareData_linear_sing.py:336-339): every tile
    is seen once per epoch.  Training: every batch has exactly `bs` tiles (the native trainer is sized for it) -- the ragged tail is filled
    up with tiles from the head of the epoch's order -- and EVERY RANK GETS THE SAME NUMBER OF BATCHES (each optimiser step is a collective:
    a rank with fewer steps would leave the others waiting in the all-reduce), the last round being filled by wrapping around.
    Validation (no collective inside the loop, no fixed batch size): plain round-robin, ragged tail kept."""
    if n <= 0:
        return []
    nb = -(-n // bs)
    if not train:
        return [list(range(i * bs, min((i + 1) * bs, n))) for i in range(rank, nb, world)]
    rounds = -(-nb // world)
    return [[j % n for j in range((i % nb) * bs, (i % nb) * bs + bs)] for i in range(rank, rounds * world, world)]


def _batches(tiles, epoch, split, rank, world, device):
    lq, hq = tiles.split(epoch, split)
    for idx in batch_plan(lq.shape[0], tiles.args.batch_size, rank, world, split == "train"):
        idx = torch.as_tensor(idx)
        yield lq[idx].to(device), hq[idx].to(device)


def main(argv=None):
    args = create_parser().parse_args(argv)
    conditional = not args.unspervised
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.manual_seed(args.seed)                                         # same initial weights on every rank
    if args.arch == "hicedrn":
        from hicdiff_amd.model.hicedrn_Diff import hicedrn_Diff
        net = hicedrn_Diff(number_resnet=args.resnet_blocks, self_condition=conditional)
    else:
        from hicdiff_amd.hicdiff import Unet
        net = Unet(64, dim_mults=(1, 2, 4, 8), self_condition=conditional)
    net.train_precision = args.precision
    optimise = not args.eval_only
    if optimise and not getattr(net, "_native_train", False):
        raise NotImplementedError("this network has no native training step; run with --eval-only to evaluate the objective")
    # HICDIFF_DEVICE / HICDIFF_DIST_BACKEND: rehearsal of the N-rank path on a one-GPU box (every rank on device 0, gloo)
    device = torch.device("cuda", int(os.environ.get("HICDIFF_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("HICDIFF_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": device} if backend == "nccl" else {}))
    if conditional:
        from hicdiff_amd.hicdiff_condition import GaussianDiffusion
    else:
        from hicdiff_amd.hicdiff import GaussianDiffusion
    from hicdiff_amd.optim import Adam
    # train.py:86-107: 1000 steps, linear schedule, l2
    diffusion = GaussianDiffusion(net, image_size=args.tile, timesteps=1000, loss_type="l2", beta_schedule="linear").to(device)
    optimizer = Adam(diffusion.parameters(), lr=args.lr)                 # train.py:111
    torch.manual_seed(args.seed + 1000 * (rank + 1))                     # timesteps and noise differ per rank
    best = float("inf")
    os.makedirs(args.weights_dir, exist_ok=True)
    # train.py:185,189 names every checkpoint `..._HiCedrn_cond_l2_lin.pytorch` with c{chunk}_s{chunk}, whatever -u says (inference.py loads
    # exactly that name); the UNet stems are those of pretrain/train_unet_Diff_cond.py:145,149
    tag = "HiCedrn_cond_l2_lin" if args.arch == "hicedrn" else "unet_cond_l2_lin"
    stem = f"g_40000_c{args.tile}_s{args.tile}_{args.celline}{args.celln}_{tag}.pytorch"
    tiles = _Tiles(args, rank, dist)

    def mean_over_ranks(total, count):
        if dist is not None:
            v = torch.tensor([total, count], dtype=torch.float64, device=device)
            dist.all_reduce(v)
            total, count = float(v[0]), float(v[1])
        return total / max(count, 1.0)

    for epoch in range(1, args.epoch + 1):
        diffusion.train()
        tot, n = 0.0, 0
        for data, target in _batches(tiles, epoch, "train", rank, world, device):
            x = [data, target] if conditional else target               # train.py:127-130
            if optimise:
                loss = diffusion(x)
                loss.backward()
                optimizer.step()
                optimizer.zero_grad()
            else:
                with torch.no_grad():
                    loss = diffusion(x)
            tot += float(loss.detach()) * data.shape[0]
            n += data.shape[0]
        train_loss = mean_over_ranks(tot, n)
        diffusion.eval()
        tot, n = 0.0, 0
        with torch.no_grad():
            for data, target in _batches(tiles, epoch, "valid", rank, world, device):
                tot += float(diffusion([data, target] if conditional else target)) * data.shape[0]
                n += data.shape[0]
        valid_loss = mean_over_ranks(tot, n)
        if rank == 0:
            print(json.dumps({"Epoch": epoch, "train/loss": train_loss, "valid/loss": valid_loss}), flush=True)
            if valid_loss < best:                                        # train.py:182-186
                best = valid_loss
                torch.save(diffusion.state_dict(), os.path.join(args.weights_dir, "best" + stem))
    if rank == 0:
        torch.save(diffusion.state_dict(), os.path.join(args.weights_dir, "final" + stem))      # train.py:189-190
    if args.print_checksum:
        flat = torch.cat([p.detach().reshape(-1).double() for p in diffusion.parameters()])
        line = json.dumps({"rank": rank, "param_sum": float(flat.sum()), "param_abs_sum": float(flat.abs().sum())}) + "\n"
        sys.stdout.flush()
        os.write(sys.stdout.fileno(), line.encode())        # one write: ranks share the launcher's pipe, and a line must not interleave with another rank's
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return best


if __name__ == "__main__":
    main()
