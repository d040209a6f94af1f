"""Live re-check of the oracle against the imported reference (build container only; skipped where
/root/reference does not exist, e.g. on the GPU box).  Fresh seeded inputs, not the fixtures."""
import os
import sys
import types

import pytest
import torch

REF = os.environ.get("HICDIFF_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference checkout not present")


@pytest.fixture(scope="module")
def ref():
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
    sys.modules.setdefault("torchvision.utils", types.ModuleType("torchvision.utils"))
    from src import hicdiff, hicdiff_condition
    from src.model import hicedrn_Diff
    return types.SimpleNamespace(hicdiff=hicdiff, cond=hicdiff_condition, hicedrn=hicedrn_Diff)


def test_unet_forward_matches_reference_on_fresh_inputs(ref):
    from oracle import nets as ON, weights as W
    torch.manual_seed(3)
    m = ref.hicdiff.Unet(dim=32, dim_mults=(1, 2, 4), channels=1).eval()
    W.fill_module_(m)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x, t = torch.randn(3, 1, 24, 24), torch.tensor([0, 13, 999])
    with torch.no_grad():
        want = m(x, t)
        got = ON.unet_eps(sd, x, t, None, ON.UnetCfg(dim=32, dim_mults=(1, 2, 4)))
    assert (want - got).abs().max() <= 1e-5 * want.abs().max()


def test_hicedrn_and_conditional_chain_match_reference(ref):
    from oracle import diffusion as OD, nets as ON, weights as W
    m = ref.hicedrn.hicedrn_Diff(number_resnet=2, self_condition=True).eval()
    W.fill_module_(m)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    cfg = ON.HicedrnCfg(number_resnet=2, self_condition=True)
    d = ref.cond.GaussianDiffusion(m, image_size=16, timesteps=30, loss_type="l2", beta_schedule="linear")
    lq = torch.rand(2, 1, 16, 16) * 2 - 1
    torch.manual_seed(11)
    with torch.no_grad():
        want = d.super_resolution(lq)
    mine = OD.DiffusionRef(ON.make_eps_fn(sd, cfg), image_size=16, timesteps=30, beta_schedule="linear", loss_type="l2", kind="cond")
    got = mine.p_sample_loop(lq, OD.TorchNoise(11))
    assert (want - got).abs().max() <= 1e-4 * want.abs().max()
