"""Closed-form, checkpoint-free weights for parity work (test infrastructure).

Real HiCDiff checkpoints are not shipped and 36-38 M random parameters are too
big to commit, so every parity fixture uses a deterministic fill keyed by the
state-dict key and the flat element index.  The same
fill is applied to the imported reference (through ``load_state_dict``) when
the golden vectors are generated and to this repo's modules in the tests, so
fixtures only need to hold inputs and outputs.

The generator is integer arithmetic (FNV-1a of the key, splitmix64 of the
index) evaluated in numpy uint64 and mapped to float64 before the cast to
float32, so it is bit-reproducible on any host.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(text: str) -> int:
    h = 0xCBF29CE484222325
    for ch in text.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def unit_noise(key: str, n: int) -> np.ndarray:
    """n float64 values in [-1, 1), a pure function of (key, index)."""
    seed = np.uint64(_fnv1a64(key))
    with np.errstate(over="ignore"):
        idx = (np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + seed) & _MASK
    z = _splitmix64(idx)
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return 2.0 * u - 1.0


def fill_tensor(key: str, shape) -> torch.Tensor:
    """Value rule per parameter kind:

    * ``*.g`` (channel LayerNorm gain) and 1-D ``*.weight`` (GroupNorm gamma): 1 + 0.2 u
    * ``*.bias``: 0.2 u
    * any other ``*.weight`` (conv / linear): u * sqrt(3 / fan_in)  (unit-variance preserving)
    * ``*.weights`` (learned sinusoidal frequencies): u
    """
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    u = unit_noise(key, n)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "g" or (leaf == "weight" and len(shape) == 1):
        v = 1.0 + 0.2 * u
    elif leaf == "bias":
        v = 0.2 * u
    elif leaf == "weight":
        fan_in = int(np.prod(shape[1:]))
        v = u * math.sqrt(3.0 / fan_in)
    else:
        v = u
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def fill_state_dict(shapes: "OrderedDict[str, tuple]") -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((k, fill_tensor(k, s)) for k, s in shapes.items())


def fill_module_(module: torch.nn.Module, prefix: str = "") -> None:
    """Overwrite every parameter of ``module`` in place with the closed-form fill.

    ``prefix`` is prepended to the parameter names before hashing so that a bare
    epsilon-network and the same network held as ``GaussianDiffusion.model`` get
    identical weights when the caller passes ``prefix=''`` for both.
    """
    with torch.no_grad():
        for name, p in module.named_parameters():
            p.copy_(fill_tensor(prefix + name, p.shape).to(p.device))


# --------------------------------------------------------------------------------------
# parameter inventories (names + shapes) of the two epsilon-networks, derived from the
# constructors at src/hicdiff.py:255-343, src/hicdiff_sr3.py:235-251,315-404,
# src/model/hicedrn_Diff.py:182-262 and src/model/hicedrn_sr3_Diff.py:245-324.
# --------------------------------------------------------------------------------------

def _resblock_shapes(out, p, cin, cout, time_dim, sr3, groups_norm=True):
    if sr3:
        out[f"{p}.noise_func.noise_func.0.weight"] = (cout, time_dim)
        out[f"{p}.noise_func.noise_func.0.bias"] = (cout,)
    else:
        out[f"{p}.mlp.1.weight"] = (2 * cout, time_dim)
        out[f"{p}.mlp.1.bias"] = (2 * cout,)
    for blk, ci in (("block1", cin), ("block2", cout)):
        out[f"{p}.{blk}.proj.weight"] = (cout, ci, 3, 3)
        out[f"{p}.{blk}.proj.bias"] = (cout,)
        out[f"{p}.{blk}.norm.weight"] = (cout,)
        out[f"{p}.{blk}.norm.bias"] = (cout,)
    if cin != cout:
        out[f"{p}.res_conv.weight"] = (cout, cin, 1, 1)
        out[f"{p}.res_conv.bias"] = (cout,)


def _linattn_shapes(out, p, dim, heads=4, dim_head=32):
    hid = heads * dim_head
    out[f"{p}.fn.fn.to_qkv.weight"] = (3 * hid, dim, 1, 1)
    out[f"{p}.fn.fn.to_out.0.weight"] = (dim, hid, 1, 1)
    out[f"{p}.fn.fn.to_out.0.bias"] = (dim,)
    out[f"{p}.fn.fn.to_out.1.g"] = (1, dim, 1, 1)
    out[f"{p}.fn.norm.g"] = (1, dim, 1, 1)


def unet_shapes(dim=64, dim_mults=(1, 2, 4, 8), channels=1, self_condition=False,
                sr3=False, init_dim=None, out_dim=None):
    """Key order follows nn.Module registration order of the reference constructor."""
    out = OrderedDict()
    init_dim = init_dim or dim
    cin0 = channels * (2 if self_condition else 1)
    out["init_conv.weight"] = (init_dim, cin0, 7, 7)
    out["init_conv.bias"] = (init_dim,)
    time_dim = dim * 4
    out["time_mlp.1.weight"] = (time_dim, dim)
    out["time_mlp.1.bias"] = (time_dim,)
    out["time_mlp.3.weight"] = (time_dim, time_dim)
    out["time_mlp.3.bias"] = (time_dim,)
    dims = [init_dim] + [dim * m for m in dim_mults]
    in_out = list(zip(dims[:-1], dims[1:]))
    n = len(in_out)
    for i, (di, do) in enumerate(in_out):
        last = i >= n - 1
        _resblock_shapes(out, f"downs.{i}.0", di, di, time_dim, sr3)
        _resblock_shapes(out, f"downs.{i}.1", di, di, time_dim, sr3)
        _linattn_shapes(out, f"downs.{i}.2", di)
        if last:
            out[f"downs.{i}.3.weight"] = (do, di, 3, 3)
            out[f"downs.{i}.3.bias"] = (do,)
        else:
            out[f"downs.{i}.3.1.weight"] = (do, di * 4, 1, 1)
            out[f"downs.{i}.3.1.bias"] = (do,)
    for i, (di, do) in enumerate(reversed(in_out)):
        last = i == n - 1
        _resblock_shapes(out, f"ups.{i}.0", do + di, do, time_dim, sr3)
        _resblock_shapes(out, f"ups.{i}.1", do + di, do, time_dim, sr3)
        _linattn_shapes(out, f"ups.{i}.2", do)
        if last:
            out[f"ups.{i}.3.weight"] = (di, do, 3, 3)
            out[f"ups.{i}.3.bias"] = (di,)
        else:
            out[f"ups.{i}.3.1.weight"] = (di, do, 3, 3)
            out[f"ups.{i}.3.1.bias"] = (di,)
    # nn.ModuleList attributes (downs, ups) are registered before the mid blocks upstream
    mid = dims[-1]
    _resblock_shapes(out, "mid_block1", mid, mid, time_dim, sr3)
    out["mid_attn.fn.fn.to_qkv.weight"] = (384, mid, 1, 1)
    out["mid_attn.fn.fn.to_out.weight"] = (mid, 128, 1, 1)
    out["mid_attn.fn.fn.to_out.bias"] = (mid,)
    out["mid_attn.fn.norm.g"] = (1, mid, 1, 1)
    _resblock_shapes(out, "mid_block2", mid, mid, time_dim, sr3)
    _resblock_shapes(out, "final_res_block", dim * 2, dim, time_dim, sr3)
    od = out_dim or channels
    out["final_conv.weight"] = (od, dim, 1, 1)
    out["final_conv.bias"] = (od,)
    return out


def hicedrn_shapes(channels=1, number_resnet=32, self_condition=False, sr3=False,
                   n_feat=256, out_dim=None):
    out = OrderedDict()
    cin0 = channels * (2 if self_condition else 1)
    out["head.weight"] = (n_feat, cin0, 3, 3)
    out["head.bias"] = (n_feat,)
    time_dim = n_feat * 4
    out["time_mlp.1.weight"] = (time_dim, n_feat)
    out["time_mlp.1.bias"] = (time_dim,)
    out["time_mlp.3.weight"] = (time_dim, time_dim)
    out["time_mlp.3.bias"] = (time_dim,)
    for i in range(number_resnet):
        if sr3:
            out[f"body.{i}.noise_func.noise_func.0.weight"] = (n_feat, time_dim)
            out[f"body.{i}.noise_func.noise_func.0.bias"] = (n_feat,)
        else:
            out[f"body.{i}.mlp.1.weight"] = (2 * n_feat, time_dim)
            out[f"body.{i}.mlp.1.bias"] = (2 * n_feat,)
        out[f"body.{i}.conv.proj.weight"] = (n_feat, n_feat, 3, 3)
        out[f"body.{i}.conv.proj.bias"] = (n_feat,)
    out["body_tail.weight"] = (n_feat, n_feat, 3, 3)
    out["body_tail.bias"] = (n_feat,)
    od = out_dim or channels
    out["tail.weight"] = (od, n_feat, 3, 3)
    out["tail.bias"] = (od,)
    return out
