#!/usr/bin/env python3
"""CPU emulation of the 64-channel LinearAttention K/V side's two partial forms (hicdiff_amd/csrc/linattn_fused.hip): does merging a workgroup's
chunks online cost accuracy against one partial per chunk?  Both forms use the same arithmetic class -- split-bf16 x3 products with fp32
accumulation, exponentials against a chunk-level or running maximum -- and differ in where the rescaling happens.

    python3 tests/studies/linattn_merge_error_study.py        (seconds; numpy + torch for the bf16 rounding)

Prints, per seed, the relative error (max |d| / max |ref|) of the normalised 32 x 32 context against float64 for: per-chunk partials + combine
(the round-3 form), online merge in groups of 8 chunks + combine over the groups (round 4), and one fp32 pass without any splitting.
"""
import numpy as np
import torch


def bf16(x):
    return torch.from_numpy(x.astype(np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def split(x):
    hi = bf16(x)
    return hi, bf16(x.astype(np.float32) - hi)


def x3(p, v):
    """sum over tokens of p[t, d] * v[t, e] as pl vh + ph vl + ph vh with fp32 accumulation"""
    ph, pl = split(p)
    vh, vl = split(v)
    f = np.float32
    return (pl.T.astype(f) @ vh.astype(f) + ph.T.astype(f) @ vl.astype(f)).astype(f) + (ph.T.astype(f) @ vh.astype(f)).astype(f)


def run(seed, HW=4096, D=32, tok=64, cpw=8, scale=2.0):
    rng = np.random.default_rng(seed)
    k = (rng.standard_normal((HW, D)) * scale).astype(np.float32)
    v = rng.standard_normal((HW, D)).astype(np.float32)
    k64, v64 = k.astype(np.float64), v.astype(np.float64)
    w = np.exp(k64 - k64.max(0))
    ref = (w / w.sum(0)).T @ v64 / HW
    f = np.float32
    # (a) one partial per chunk, two-level combine
    parts = []
    for n0 in range(0, HW, tok):
        kc, vc = k[n0:n0 + tok], v[n0:n0 + tok]
        mx = kc.max(0)
        p = np.exp((kc - mx).astype(f)).astype(f)
        parts.append((mx, p.sum(0, dtype=f), x3(p, vc)))
    def combine(parts):
        gm = np.max([m for m, _, _ in parts], 0)
        s = np.zeros(D, f); a = np.zeros((D, D), f)
        for m, ps, c in parts:
            fac = np.exp((m - gm).astype(f)).astype(f)
            s = (s + ps * fac).astype(f); a = (a + c * fac[:, None]).astype(f)
        return a / (s[:, None] * f(HW))
    per_chunk = combine(parts)
    # (b) online merge inside groups of cpw chunks, then the same combine over the groups
    groups = []
    nchunks = HW // tok
    for g0 in range(0, nchunks, cpw):
        m_run = np.full(D, -3.0e38, f); s_run = np.zeros(D, f); ctx = np.zeros((D, D), f)
        for c in range(g0, min(nchunks, g0 + cpw)):
            kc, vc = k[c * tok:(c + 1) * tok], v[c * tok:(c + 1) * tok]
            m_new = np.maximum(m_run, kc.max(0))
            f_old = np.exp((m_run - m_new).astype(f)).astype(f)
            s_run = (s_run * f_old).astype(f); ctx = (ctx * f_old[:, None]).astype(f)
            p = np.exp((kc - m_new).astype(f)).astype(f)
            s_run = (s_run + p.sum(0, dtype=f)).astype(f); ctx = (ctx + x3(p, vc)).astype(f)
            m_run = m_new
        groups.append((m_run, s_run, ctx))
    merged = combine(groups)
    # (c) plain fp32
    p = np.exp((k - k.max(0)).astype(f)).astype(f)
    plain = (p.T @ v) / (p.sum(0, dtype=f)[:, None] * f(HW))
    err = lambda x: np.abs(x.astype(np.float64) - ref).max() / np.abs(ref).max()
    return err(per_chunk), err(merged), err(plain)


if __name__ == "__main__":
    print("seed   per-chunk partials   merged in groups of 8   plain fp32")
    rows = [run(s) for s in range(8)]
    for s, (a, b, c) in enumerate(rows):
        print(f"{s:4d}   {a:.3e}            {b:.3e}               {c:.3e}")
    m = np.mean(rows, 0)
    print(f"mean   {m[0]:.3e}            {m[1]:.3e}               {m[2]:.3e}")
