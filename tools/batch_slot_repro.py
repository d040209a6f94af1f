#!/usr/bin/env python3
"""Checks that a tile's result does not depend on its position in a two-image workgroup tile (a loop form whose
`s += x * x` hipcc contracted in some instances and not in others made the lower-half samples of the 8x8 maps differ by 4e-5 from the same
samples computed alone).  Every figure printed must be 0.  The reduced form of the effect: tools/epilogue_sum_repro.hip."""
import os
import sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _util import product_unet, tiles
m = product_unet("uncond")
x = tiles(3, 64, 64).cuda()
t = torch.randint(0, 1000, (64,), generator=torch.Generator().manual_seed(1)).cuda()
full = m(x, t)
part = torch.cat([m(x[:17], t[:17]), m(x[17:], t[17:])])
d = (full - part).abs().flatten(1).max(dim=1).values
print("odd split: samples that differ:", int((d > 0).sum()), "max", float(d.max()))
one = torch.cat([m(x[i:i + 1], t[i:i + 1]) for i in (4, 5)])
print("samples 4 (lower half) / 5 (upper half) alone vs in batch:", float((one[0] - full[4]).abs().max()), float((one[1] - full[5]).abs().max()))
