"""Shared helpers of the test-suite: golden fixtures, closed-form weights, product/oracle builders."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


def rel_err(ref, got):
    ref, got = ref.detach().float().cpu(), got.detach().float().cpu()
    return ((ref - got).abs().max() / ref.abs().max().clamp_min(1e-12)).item()


def fill_product_(module, device="cuda"):
    """Closed-form weights (oracle/weights.py) pushed into a hicdiff_amd module, then moved to the GPU."""
    from oracle import weights as W
    W.fill_module_(module)
    return module.to(device)


def oracle_unet(kind, dim=64, mults=(1, 2, 4, 8)):
    from oracle import nets as ON, weights as W
    cfg = ON.UnetCfg(dim=dim, dim_mults=tuple(mults), self_condition=(kind != "uncond"), sr3=(kind == "sr3"))
    sd = W.fill_state_dict(W.unet_shapes(dim=dim, dim_mults=tuple(mults), self_condition=cfg.self_condition, sr3=cfg.sr3))
    return ON.make_eps_fn(sd, cfg)


def oracle_hicedrn(kind, nres):
    from oracle import nets as ON, weights as W
    cfg = ON.HicedrnCfg(number_resnet=nres, self_condition=(kind != "uncond"), sr3=(kind == "sr3"))
    sd = W.fill_state_dict(W.hicedrn_shapes(number_resnet=nres, self_condition=cfg.self_condition, sr3=cfg.sr3))
    return ON.make_eps_fn(sd, cfg)


def product_unet(kind, dim=64, mults=(1, 2, 4, 8), device="cuda"):
    if kind == "uncond":
        from hicdiff_amd.hicdiff import Unet
        m = Unet(dim, dim_mults=mults, self_condition=False)
    elif kind == "cond":
        from hicdiff_amd.hicdiff_condition import Unet
        m = Unet(dim, dim_mults=mults, self_condition=True)
    else:
        from hicdiff_amd.hicdiff_sr3 import Unet
        m = Unet(dim, dim_mults=mults, self_condition=True, noise_level_emb=True)
    return fill_product_(m, device)


def product_hicedrn(kind, nres, device="cuda"):
    if kind == "sr3":
        from hicdiff_amd.model.hicedrn_sr3_Diff import hicedrn_Diff
        m = hicedrn_Diff(number_resnet=nres, self_condition=True, noise_level_emb=True)
    else:
        from hicdiff_amd.model.hicedrn_Diff import hicedrn_Diff
        m = hicedrn_Diff(number_resnet=nres, self_condition=(kind == "cond"))
    return fill_product_(m, device)


def diffusion_class(kind):
    if kind == "uncond":
        from hicdiff_amd.hicdiff import GaussianDiffusion
    elif kind == "cond":
        from hicdiff_amd.hicdiff_condition import GaussianDiffusion
    else:
        from hicdiff_amd.hicdiff_sr3 import GaussianDiffusion
    return GaussianDiffusion


def tiles(seed, b, s, c=1):
    g = torch.Generator().manual_seed(seed)
    return torch.rand((b, c, s, s), generator=g) * 2 - 1
