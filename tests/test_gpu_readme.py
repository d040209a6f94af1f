"""README.md's usage blocks, executed as they stand (the round-3 snippet called sample(batch_size=...) and raised TypeError)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def python_blocks():
    doc = open(os.path.join(ROOT, "README.md")).read()
    sec = doc[doc.index("Usage is the reference's"):]
    return re.findall(r"```python\n(.*?)```", sec, re.S)


def test_readme_blocks_compile_and_use_the_reference_signatures():
    blocks = python_blocks()
    assert len(blocks) >= 2
    for i, b in enumerate(blocks):
        compile(b, f"README.md:block{i}", "exec")
    assert "diffusion.sample(torch.zeros(256, 1, 64, 64))" in blocks[0]          # sample(x): src/hicdiff.py:667
    assert "loss.backward(); optimizer.step(); optimizer.zero_grad()" in blocks[1]    # train.py:131-134


@pytest.mark.gpu
def test_readme_blocks_run(tmp_path):
    from hicdiff_amd.hicdiff import GaussianDiffusion, Unet
    blocks = python_blocks()
    # a checkpoint in the reference's form: torch.save(diffusion.state_dict()) (train.py:186)
    torch.manual_seed(0)
    d0 = GaussianDiffusion(Unet(64, dim_mults=(1, 2, 4, 8)), image_size=64, timesteps=1000, loss_type="l2")
    ckpt = str(tmp_path / "bestg_40000_c64_s64_Human1_deno_0.1_hicedrn_l2.pytorch")
    torch.save(d0.state_dict(), ckpt)
    ns = {"ckpt": ckpt}
    exec(compile(blocks[0], "README.md:block0", "exec"), ns)
    tiles = ns["tiles"]
    assert tuple(tiles.shape) == (256, 1, 64, 64) and bool(torch.isfinite(tiles).all()) and float(tiles.abs().max()) <= 1.0
    g = torch.Generator().manual_seed(1)
    hq = (torch.rand((8, 1, 64, 64), generator=g) * 2 - 1).cuda()
    lq = (hq + 0.1 * torch.randn(hq.shape, generator=g).cuda()).clamp(-1, 1)
    ns2 = {"lq": lq, "hq": hq}
    exec(compile(blocks[1], "README.md:block1", "exec"), ns2)
    assert float(ns2["loss"].detach()) > 0 and all(p.grad is None or bool(torch.isfinite(p.grad).all()) for p in ns2["diffusion"].parameters())
