#!/usr/bin/env python3
"""Training driver with the reference's command line (``train.py`` :28-42) on the MI355X engine.

Scope note: the hot path built so far is sampling.  The engine evaluates the
training objective ``GaussianDiffusion.forward`` (t ~ U{0..T-1}, q_sample, epsilon-network, l1|l2 loss;
src/hicdiff.py:711-755) but has no backward kernels yet, so this driver runs the reference's epoch
loop as a LOSS-EVALUATION loop over synthetic tiles (train and validation splits), logs
``Epoch / train/loss / valid/loss`` as JSON lines (the reference logs the same keys to wandb,
train.py:187) and writes the checkpoint under the reference's file name.  ``--optimize`` (the Adam step of
train.py:133-135) raises until the backward path exists.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def create_parser():
    p = argparse.ArgumentParser(description="HiCDiff training objective on MI355X")
    p.add_argument("-u", "--unspervised", type=bool, default=True)       # reference semantics: '' -> conditional
    p.add_argument("-b", "--batch_size", type=int, default=64)
    p.add_argument("-e", "--epoch", type=int, default=400)
    p.add_argument("-l", "--celline", type=str, default="Human")
    p.add_argument("-n", "--celln", type=int, default=1)
    p.add_argument("-s", "--sigma", type=float, default=0.1)
    p.add_argument("--arch", choices=["hicedrn", "unet"], default="hicedrn")
    p.add_argument("--resnet-blocks", type=int, default=32)
    p.add_argument("--tile", type=int, default=64)
    p.add_argument("--tiles-per-epoch", type=int, default=256)
    p.add_argument("--optimize", action="store_true", help="run the Adam step (needs backward kernels: not built yet)")
    p.add_argument("--weights-dir", default=os.path.join(ROOT, "Model_Weights"))
    p.add_argument("--seed", type=int, default=1234)
    return p


def main(argv=None):
    args = create_parser().parse_args(argv)
    if args.optimize:
        raise NotImplementedError("the optimiser step needs the backward kernels (SURVEY.md section 8 row f-2); "
                                  "this build evaluates the training objective only")
    conditional = not args.unspervised
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)
    torch.manual_seed(args.seed)
    from inference import synthetic_tiles
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
    # train.py:86-107: 1000 steps, linear schedule, l2
    diffusion = GaussianDiffusion(net, image_size=args.tile, timesteps=1000, loss_type="l2", beta_schedule="linear").to(device)
    best = float("inf")
    os.makedirs(args.weights_dir, exist_ok=True)
    tag = "HiCedrn" if args.arch == "hicedrn" else "Unet"
    name = f"bestg_40000_c64_s{args.tile}_{args.celline}{args.celln}_{tag}{'_cond' if conditional else ''}_l2_lin.pytorch"
    for epoch in range(args.epoch):
        sums = {}
        for split, seed in (("train", args.seed + 2 * epoch), ("valid", args.seed + 2 * epoch + 1)):
            lq, hq = synthetic_tiles(args.tiles_per_epoch, args.tile, args.sigma, seed)
            tot, nb = 0.0, 0
            for b0 in range(0, lq.shape[0], args.batch_size):
                data, target = lq[b0:b0 + args.batch_size].to(device), hq[b0:b0 + args.batch_size].to(device)
                x = [data, target] if conditional else target          # train.py:127-130
                with torch.no_grad():
                    tot += float(diffusion(x))
                nb += 1
            sums[split] = tot / max(nb, 1)
        print(json.dumps({"Epoch": epoch, "train/loss": sums["train"], "valid/loss": sums["valid"]}), flush=True)
        if sums["valid"] < best:                                       # train.py:182-186
            best = sums["valid"]
            torch.save(diffusion.state_dict(), os.path.join(args.weights_dir, name))
    return best


if __name__ == "__main__":
    main()
