"""INTEGRATION.md's Level-2 binding is executed as written: the fenced code block is cut out of the document and run
against the library, so the documented stub cannot drift from include/hicdiff_hip.h (round 2 shipped a 6-float Coef
next to a 7-float hd_ddpm_coef).  The stub's p_sample_inplace must equal the mirror's p_sample
(hicdiff_amd/_diffusion.py, the stand-in for /root/reference/src/hicdiff.py:594-601) bit for bit; a struct of a size the
library does not know must be refused with HD_EINVAL instead of being read past its end."""
import ctypes as C
import os
import re

import pytest
import torch

from _util import fill_product_, tiles

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def level2_block() -> str:
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("## Level 2"):]
    m = re.search(r"```python\n(.*?)```", sec, re.S)
    assert m, "INTEGRATION.md: no fenced python block under '## Level 2'"
    return m.group(1)


def test_level2_block_matches_the_binding():
    """CPU: the documented structs have the sizes / field order of hicdiff_amd/_lib.py (which load() checks against the library)."""
    src = level2_block()
    compile(src, "INTEGRATION.md:Level-2", "exec")
    from hicdiff_amd import _lib as L
    # the struct definitions of the block do not touch the library: run just those
    ns = {"C": C}
    for m in re.finditer(r"^class (\w+)\(C\.Structure\):.*?(?=^\S)", src, re.S | re.M):
        exec(m.group(0), ns)
    assert C.sizeof(ns["Coef"]) == C.sizeof(L.HdDdpmCoef) == 36
    assert [f[0] for f in ns["Coef"]._fields_] == [f[0] for f in L.HdDdpmCoef._fields_]
    assert C.sizeof(ns["Arch"]) == C.sizeof(L.HdArchDesc)
    assert C.sizeof(ns["Named"]) == C.sizeof(L.HdNamedTensor)
    assert f"HD_ABI_VERSION = {L.HD_ABI_VERSION}" in src
    hdr = open(os.path.join(ROOT, "include", "hicdiff_hip.h")).read()
    assert f"#define HD_ABI_VERSION {L.HD_ABI_VERSION}" in hdr


@pytest.mark.gpu
def test_level2_stub_runs_and_equals_the_mirror():
    from hicdiff_amd import _lib as L
    from hicdiff_amd.hicdiff import GaussianDiffusion, HostReplayNoise, Unet

    dev = torch.device("cuda", 0)
    dim, mults, S, B, T = 16, (1, 2), 16, 3, 50
    net = fill_product_(Unet(dim, dim_mults=mults))
    diff = GaussianDiffusion(net, image_size=S, timesteps=T, loss_type="l2", beta_schedule="linear").to(dev)

    os.environ["HICDIFF_HIP_LIB"] = L.LIB_PATH
    ns = {}
    exec(compile(level2_block(), "INTEGRATION.md:Level-2", "exec"), ns)
    ctx = ns["make_ctx"](net, dim=dim, dim_mults=mults)
    lib = ns["lib"]
    try:
        x = tiles(3, B, S).to(dev)
        for t in (T - 1, 17, 0):
            noise = torch.randn(B, 1, S, S, generator=torch.Generator().manual_seed(t)).to(dev) if t > 0 else None
            diff.noise_source = None if noise is None else type("N", (), {"randn": staticmethod(lambda shape, n=noise: n)})()
            want, _ = diff.p_sample(x, t)
            got = x.clone()
            ns["p_sample_inplace"](diff, ctx, got, t, None, noise)
            torch.cuda.synchronize()
            assert torch.equal(got, want), f"documented stub != mirror at t={t}: {(got - want).abs().max().item()}"
            assert not torch.equal(got, x)
            x = want

        # a pre-revision-3 binding (no struct_bytes: the first field is a float) and a truncated struct are refused, not read past
        Coef = ns["Coef"]
        img = x.clone()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

        class OldCoef(C.Structure):
            _fields_ = [(n, C.c_float) for n in ("a", "b", "c", "d", "e", "f")]

        for bad in (OldCoef(1.2, 0.3, 0.5, 0.5, 0.1, 7.0), Coef(struct_bytes=12), Coef(struct_bytes=64), Coef(struct_bytes=30)):
            rc = lib.hd_ddpm_step(ctx, C.c_void_p(img.data_ptr()), None, None, C.byref(bad), None, B, S, C.c_uint64(1), C.c_uint64(0), C.c_uint32(3), stream)
            assert rc == L.HD_EINVAL, rc
            assert b"struct_bytes" in lib.hd_last_error(ctx)
        torch.cuda.synchronize()
        assert torch.equal(img, x), "a refused call must not touch the tile"

        # an older revision-3 struct without eps_coef (28 bytes) is an ancestral step
        class ShortCoef(C.Structure):
            _fields_ = Coef._fields_[:-1]
        full = Coef(struct_bytes=C.sizeof(Coef), sqrt_recip_alphas_cumprod=1.1, sqrt_recipm1_alphas_cumprod=0.4, posterior_mean_coef1=0.6,
                    posterior_mean_coef2=0.4, sigma=0.0, time_value=5.0, eps_coef=0.0)
        short = ShortCoef(struct_bytes=C.sizeof(ShortCoef), sqrt_recip_alphas_cumprod=1.1, sqrt_recipm1_alphas_cumprod=0.4, posterior_mean_coef1=0.6,
                          posterior_mean_coef2=0.4, sigma=0.0, time_value=5.0)
        outs = []
        for c in (full, short):
            y = x.clone()
            rc = lib.hd_ddpm_step(ctx, C.c_void_p(y.data_ptr()), None, None, C.byref(c), None, B, S, C.c_uint64(1), C.c_uint64(0), C.c_uint32(5), stream)
            assert rc == 0, lib.hd_last_error(ctx)
            torch.cuda.synchronize()
            outs.append(y)
        assert torch.equal(outs[0], outs[1])
    finally:
        lib.hd_destroy(ctx)
