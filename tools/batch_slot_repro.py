#!/usr/bin/env python3
"""Repro driver for the two-image-tile asymmetry of DESIGN.md section 8 (GPU box):
    cd hicdiff_amd/csrc && rm -f conv_bf16x3_ck32.o conv_bf16x3_ck16.o && make EXTRA=-DHD_EPI_V9 && cd ../.. && python3 tools/batch_slot_repro.py
With the product build every figure printed is 0; with HD_EPI_V9 the samples whose 8x8 maps sit in the LOWER half of a two-image tile differ
from the same samples computed alone (4e-5), the upper-half ones do not."""
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
