"""fp32 CPU restatement of the two HiCDiff noise predictors (test infrastructure).

Functional style: every network is a plain function of (state_dict, inputs); no
nn.Module mirrors the reference classes.  Citations are upstream ``file:line``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Sequence

import torch
import torch.nn.functional as F

EPS = 1e-5  # fp32 branch of the reference's eps selection (src/hicdiff.py:90,105)


@dataclass(frozen=True)
class UnetCfg:
    dim: int = 64
    dim_mults: Sequence[int] = (1, 2, 4, 8)
    channels: int = 1
    self_condition: bool = False
    sr3: bool = False            # noise_level_emb=True flavour (src/hicdiff_sr3.py)
    groups: int = 8
    heads: int = 4
    dim_head: int = 32


@dataclass(frozen=True)
class HicedrnCfg:
    channels: int = 1
    number_resnet: int = 32
    self_condition: bool = False
    sr3: bool = False
    n_feat: int = 256


# ---------------------------------------------------------------- time embeddings

def sinusoidal_pos_emb(t: torch.Tensor, dim: int) -> torch.Tensor:
    """src/hicdiff.py:122-134: divisor is half_dim - 1; output [sin | cos]."""
    half = dim // 2
    k = math.log(10000) / (half - 1)
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * -k)
    arg = t.float()[:, None] * freq[None, :]
    return torch.cat((arg.sin(), arg.cos()), dim=-1)


def noise_level_encoding(level: torch.Tensor, dim: int) -> torch.Tensor:
    """src/hicdiff_sr3.py:155-165: step = k / count (not count - 1); input (B,1) or (B,)."""
    count = dim // 2
    step = torch.arange(count, dtype=level.dtype) / count
    arg = level.reshape(-1, 1) * torch.exp(-math.log(1e4) * step)[None, :]
    return torch.cat((arg.sin(), arg.cos()), dim=-1)


def time_mlp(sd, t, dim, sr3):
    """Linear -> exact (erf) GELU -> Linear, src/hicdiff.py:300-305."""
    e = noise_level_encoding(t.float(), dim) if sr3 else sinusoidal_pos_emb(t, dim)
    e = F.linear(e, sd["time_mlp.1.weight"], sd["time_mlp.1.bias"])
    e = F.gelu(e)
    return F.linear(e, sd["time_mlp.3.weight"], sd["time_mlp.3.bias"])


# ---------------------------------------------------------------- UNet pieces

def ws_conv3x3(sd, p, x):
    """Weight-standardised conv, biased variance, eps inside rsqrt (src/hicdiff.py:84-97)."""
    w = sd[p + ".weight"]
    mean = w.mean(dim=(1, 2, 3), keepdim=True)
    var = w.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
    return F.conv2d(x, (w - mean) * (var + EPS).rsqrt(), sd[p + ".bias"], padding=1)


def gn_block(sd, p, x, groups, scale_shift=None):
    """Block.forward src/hicdiff.py:162-171: WS-conv -> GroupNorm -> FiLM -> SiLU."""
    x = ws_conv3x3(sd, p + ".proj", x)
    x = F.group_norm(x, groups, sd[p + ".norm.weight"], sd[p + ".norm.bias"], eps=1e-5)
    if scale_shift is not None:
        scale, shift = scale_shift
        x = x * (scale + 1) + shift
    return F.silu(x)


def unet_resblock(sd, p, x, temb, cfg: UnetCfg):
    """ResnetBlock.forward src/hicdiff.py:185-197 (FiLM on block1 only);
    SR3 flavour src/hicdiff_sr3.py:246-251 (additive noise embedding after block1)."""
    if cfg.sr3:
        h = gn_block(sd, p + ".block1", x, cfg.groups)
        e = F.linear(temb, sd[p + ".noise_func.noise_func.0.weight"], sd[p + ".noise_func.noise_func.0.bias"])
        h = h + e[:, :, None, None]
    else:
        e = F.linear(F.silu(temb), sd[p + ".mlp.1.weight"], sd[p + ".mlp.1.bias"])
        scale, shift = e[:, :, None, None].chunk(2, dim=1)
        h = gn_block(sd, p + ".block1", x, cfg.groups, (scale, shift))
    h = gn_block(sd, p + ".block2", h, cfg.groups)
    if (p + ".res_conv.weight") in sd:
        x = F.conv2d(x, sd[p + ".res_conv.weight"], sd[p + ".res_conv.bias"])
    return h + x


def channel_layernorm(x, g):
    """src/hicdiff.py:104-108: per-pixel norm over channels, biased var, (var+eps).rsqrt()."""
    var = x.var(dim=1, unbiased=False, keepdim=True)
    mean = x.mean(dim=1, keepdim=True)
    return (x - mean) * (var + EPS).rsqrt() * g


def linear_attention(sd, p, x, cfg: UnetCfg):
    """Residual(PreNorm(LinearAttention)) src/hicdiff.py:212-227 with wrappers :64-70,110-118."""
    b, c, h, w = x.shape
    n = h * w
    y = channel_layernorm(x, sd[p + ".fn.norm.g"])
    qkv = F.conv2d(y, sd[p + ".fn.fn.to_qkv.weight"])
    q, k, v = (t.reshape(b, cfg.heads, cfg.dim_head, n) for t in qkv.chunk(3, dim=1))
    q = q.softmax(dim=-2) * cfg.dim_head ** -0.5
    k = k.softmax(dim=-1)
    v = v / n
    context = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", context, q).reshape(b, cfg.heads * cfg.dim_head, h, w)
    out = F.conv2d(out, sd[p + ".fn.fn.to_out.0.weight"], sd[p + ".fn.fn.to_out.0.bias"])
    return channel_layernorm(out, sd[p + ".fn.fn.to_out.1.g"]) + x


def full_attention(sd, p, x, cfg: UnetCfg):
    """Residual(PreNorm(Attention)) src/hicdiff.py:239-251."""
    b, c, h, w = x.shape
    n = h * w
    y = channel_layernorm(x, sd[p + ".fn.norm.g"])
    qkv = F.conv2d(y, sd[p + ".fn.fn.to_qkv.weight"])
    q, k, v = (t.reshape(b, cfg.heads, cfg.dim_head, n) for t in qkv.chunk(3, dim=1))
    sim = torch.einsum("bhdi,bhdj->bhij", q * cfg.dim_head ** -0.5, k)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bhij,bhdj->bhid", attn, v)               # (b, heads, n, d)
    out = out.permute(0, 1, 3, 2).reshape(b, cfg.heads * cfg.dim_head, h, w)
    out = F.conv2d(out, sd[p + ".fn.fn.to_out.weight"], sd[p + ".fn.fn.to_out.bias"])
    return out + x


def pixel_unshuffle_conv(sd, p, x):
    """Downsample src/hicdiff.py:78-82: 'b c (h p1) (w p2) -> b (c p1 p2) h w' then 1x1."""
    b, c, hh, ww = x.shape
    y = x.reshape(b, c, hh // 2, 2, ww // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(b, c * 4, hh // 2, ww // 2)
    return F.conv2d(y, sd[p + ".1.weight"], sd[p + ".1.bias"])


def upsample_conv(sd, p, x):
    """Upsample src/hicdiff.py:72-76: nearest x2 then 3x3."""
    y = F.interpolate(x, scale_factor=2, mode="nearest")
    return F.conv2d(y, sd[p + ".1.weight"], sd[p + ".1.bias"], padding=1)


def unet_eps(sd, x, t, cond: Optional[torch.Tensor], cfg: UnetCfg, probes: Optional[dict] = None):
    """Unet.forward src/hicdiff.py:345-387 (conditional concat order: (cond, x), :352)."""
    if cfg.self_condition:
        x = torch.cat((cond, x), dim=1)
    x = F.conv2d(x, sd["init_conv.weight"], sd["init_conv.bias"], padding=3)
    r = x
    temb = time_mlp(sd, t, cfg.dim, cfg.sr3)
    if probes is not None:
        probes["init_conv"] = x
        probes["time_mlp"] = temb
    n = len(cfg.dim_mults)
    skips = []
    for i in range(n):
        x = unet_resblock(sd, f"downs.{i}.0", x, temb, cfg)
        skips.append(x)
        x = unet_resblock(sd, f"downs.{i}.1", x, temb, cfg)
        x = linear_attention(sd, f"downs.{i}.2", x, cfg)
        skips.append(x)
        if i >= n - 1:
            x = F.conv2d(x, sd[f"downs.{i}.3.weight"], sd[f"downs.{i}.3.bias"], padding=1)
        else:
            x = pixel_unshuffle_conv(sd, f"downs.{i}.3", x)
        if probes is not None:
            probes[f"downs.{i}"] = x
    x = unet_resblock(sd, "mid_block1", x, temb, cfg)
    x = full_attention(sd, "mid_attn", x, cfg)
    x = unet_resblock(sd, "mid_block2", x, temb, cfg)
    if probes is not None:
        probes["mid"] = x
    for i in range(n):
        x = unet_resblock(sd, f"ups.{i}.0", torch.cat((x, skips.pop()), dim=1), temb, cfg)
        x = unet_resblock(sd, f"ups.{i}.1", torch.cat((x, skips.pop()), dim=1), temb, cfg)
        x = linear_attention(sd, f"ups.{i}.2", x, cfg)
        if i == n - 1:
            x = F.conv2d(x, sd[f"ups.{i}.3.weight"], sd[f"ups.{i}.3.bias"], padding=1)
        else:
            x = upsample_conv(sd, f"ups.{i}.3", x)
        if probes is not None:
            probes[f"ups.{i}"] = x
    x = unet_resblock(sd, "final_res_block", torch.cat((x, r), dim=1), temb, cfg)
    return F.conv2d(x, sd["final_conv.weight"], sd["final_conv.bias"])


# ---------------------------------------------------------------- hicedrn

def hicedrn_eps(sd, x, t, cond: Optional[torch.Tensor], cfg: HicedrnCfg, probes: Optional[dict] = None):
    """hicedrn_Diff.forward src/model/hicedrn_Diff.py:267-289; block :194-208 applies the
    SAME 3x3 conv twice; SR3 block src/model/hicedrn_sr3_Diff.py:254-265."""
    if cfg.self_condition:
        x = torch.cat((cond, x), dim=1)
    x = F.conv2d(x, sd["head.weight"], sd["head.bias"], padding=1)
    r = x
    temb = time_mlp(sd, t, cfg.n_feat, cfg.sr3)
    if probes is not None:
        probes["head"] = x
        probes["time_mlp"] = temb
    for i in range(cfg.number_resnet):
        p = f"body.{i}"
        w, bias = sd[p + ".conv.proj.weight"], sd[p + ".conv.proj.bias"]
        h = F.conv2d(x, w, bias, padding=1)
        if cfg.sr3:
            e = F.linear(temb, sd[p + ".noise_func.noise_func.0.weight"], sd[p + ".noise_func.noise_func.0.bias"])
            h = h + e[:, :, None, None]
        else:
            e = F.linear(F.silu(temb), sd[p + ".mlp.1.weight"], sd[p + ".mlp.1.bias"])
            scale, shift = e[:, :, None, None].chunk(2, dim=1)
            h = h * (scale + 1) + shift
        h = F.silu(h)
        h = F.conv2d(h, w, bias, padding=1)
        x = h * 0.1 + x
        if probes is not None and i in (0, cfg.number_resnet - 1):
            probes[f"body.{i}"] = x
    x = F.conv2d(x, sd["body_tail.weight"], sd["body_tail.bias"], padding=1) + r
    return F.conv2d(x, sd["tail.weight"], sd["tail.bias"], padding=1)


def make_eps_fn(sd, cfg):
    """Bind weights: returns model(x, t, cond=None) like the reference's nn.Module call."""
    fn = unet_eps if isinstance(cfg, UnetCfg) else hicedrn_eps

    def model(x, t, cond=None):
        with torch.no_grad():
            return fn(sd, x, t, cond, cfg)
    model.cfg = cfg
    return model
