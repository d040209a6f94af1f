"""Parameter inventories of the two epsilon-networks: (state-dict key, shape, init kind, fan_in) in
the registration order of the reference constructors (src/hicdiff.py:270-343,
src/hicdiff_sr3.py:235-251,326-404, src/model/hicedrn_Diff.py:219-262,
src/model/hicedrn_sr3_Diff.py:245-324), so ``state_dict()`` is key-for-key what the reference
writes to ``Model_Weights/*.pytorch`` (train.py:185-190)."""
from __future__ import annotations


def _conv(out, name, cout, cin, k, bias=True):
    fan = cin * k * k
    out.append((name + ".weight", (cout, cin, k, k), "uniform", fan))
    if bias:
        out.append((name + ".bias", (cout,), "uniform", fan))


def _linear(out, name, nout, nin):
    out.append((name + ".weight", (nout, nin), "uniform", nin))
    out.append((name + ".bias", (nout,), "uniform", nin))


def _unet_res(out, p, cin, cout, time_dim, sr3):
    if sr3:
        _linear(out, p + ".noise_func.noise_func.0", cout, time_dim)
    else:
        _linear(out, p + ".mlp.1", 2 * cout, time_dim)
    for blk, ci in (("block1", cin), ("block2", cout)):
        _conv(out, f"{p}.{blk}.proj", cout, ci, 3)
        out.append((f"{p}.{blk}.norm.weight", (cout,), "ones", 0))
        out.append((f"{p}.{blk}.norm.bias", (cout,), "zeros", 0))
    if cin != cout:
        _conv(out, p + ".res_conv", cout, cin, 1)


def _linattn(out, p, dim, hidden=128):
    _conv(out, p + ".fn.fn.to_qkv", 3 * hidden, dim, 1, bias=False)
    _conv(out, p + ".fn.fn.to_out.0", dim, hidden, 1)
    out.append((p + ".fn.fn.to_out.1.g", (1, dim, 1, 1), "ones", 0))
    out.append((p + ".fn.norm.g", (1, dim, 1, 1), "ones", 0))


def unet_specs(dim, dim_mults, channels, self_condition, sr3, init_dim=None, out_dim=None):
    out = []
    init_dim = init_dim or dim
    _conv(out, "init_conv", init_dim, channels * (2 if self_condition else 1), 7)
    time_dim = dim * 4
    _linear(out, "time_mlp.1", time_dim, dim)
    _linear(out, "time_mlp.3", time_dim, time_dim)
    dims = [init_dim] + [dim * m for m in dim_mults]
    pairs = list(zip(dims[:-1], dims[1:]))
    n = len(pairs)
    for i, (di, do) in enumerate(pairs):
        _unet_res(out, f"downs.{i}.0", di, di, time_dim, sr3)
        _unet_res(out, f"downs.{i}.1", di, di, time_dim, sr3)
        _linattn(out, f"downs.{i}.2", di)
        if i >= n - 1:
            _conv(out, f"downs.{i}.3", do, di, 3)
        else:
            _conv(out, f"downs.{i}.3.1", do, di * 4, 1)
    for i, (di, do) in enumerate(reversed(pairs)):
        _unet_res(out, f"ups.{i}.0", do + di, do, time_dim, sr3)
        _unet_res(out, f"ups.{i}.1", do + di, do, time_dim, sr3)
        _linattn(out, f"ups.{i}.2", do)
        _conv(out, f"ups.{i}.3" if i == n - 1 else f"ups.{i}.3.1", di, do, 3)
    mid = dims[-1]
    _unet_res(out, "mid_block1", mid, mid, time_dim, sr3)
    _conv(out, "mid_attn.fn.fn.to_qkv", 384, mid, 1, bias=False)
    _conv(out, "mid_attn.fn.fn.to_out", mid, 128, 1)
    out.append(("mid_attn.fn.norm.g", (1, mid, 1, 1), "ones", 0))
    _unet_res(out, "mid_block2", mid, mid, time_dim, sr3)
    _unet_res(out, "final_res_block", dim * 2, dim, time_dim, sr3)
    _conv(out, "final_conv", out_dim or channels, dim, 1)
    return out


def hicedrn_specs(channels, number_resnet, self_condition, sr3, n_feat=256, out_dim=None):
    out = []
    _conv(out, "head", n_feat, channels * (2 if self_condition else 1), 3)
    time_dim = n_feat * 4
    _linear(out, "time_mlp.1", time_dim, n_feat)
    _linear(out, "time_mlp.3", time_dim, time_dim)
    for i in range(number_resnet):
        if sr3:
            _linear(out, f"body.{i}.noise_func.noise_func.0", n_feat, time_dim)
        else:
            _linear(out, f"body.{i}.mlp.1", 2 * n_feat, time_dim)
        _conv(out, f"body.{i}.conv.proj", n_feat, n_feat, 3)
    _conv(out, "body_tail", n_feat, n_feat, 3)
    _conv(out, "tail", out_dim or channels, n_feat, 3)
    return out
