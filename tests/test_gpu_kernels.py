"""Kernel-level parity on the GPU: the implicit-GEMM convolution (every loader / geometry mode the
networks use) against torch CPU fp32 conv2d, through the test-only C entry point hd_debug_conv."""
import os
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from _util import rel_err

pytestmark = pytest.mark.gpu
# mode bits of hd_debug_conv selecting the arithmetic: 0 exact fp32 MFMA (differences are summation
# order only), 32 split-bf16 x3 with 16-channel K slices, 96 the same with 32-channel slices.
PRECS = {0: 2e-5, 32: 1e-4, 96: 1e-4}


@pytest.fixture(params=sorted(PRECS), ids=["f32", "bf16x3_ck16", "bf16x3_ck32"])
def prec(request):
    return request.param


def _lib():
    from hicdiff_amd import _lib as L
    lib = L.load()
    lib.hd_debug_conv.restype = C.c_int
    lib.hd_debug_conv.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def run_conv(x0, x1, w, bias, K, mode, A=None, Bv=None, E=None, out_hw=None):
    lib = _lib()
    dev = "cuda"
    B, C0, IH, IW = x0.shape
    C1 = 0 if x1 is None else x1.shape[1]
    Cout = w.shape[0]
    H, W = out_hw or (IH, IW)
    d = lambda t: None if t is None else t.to(dev).contiguous()
    x0d, x1d = d(nhwc(x0)), (None if x1 is None else d(nhwc(x1)))
    wd, bd, Ad, Bd, Ed = d(w), d(bias), d(A), d(Bv), d(E)
    out = torch.full((B, H, W, Cout), float("nan"), device=dev)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    rc = lib.hd_debug_conv(p(x0d), C0, p(x1d), C1, B, IH, IW, p(wd), p(bd), Cout, K, mode, p(Ad), p(Bd), p(Ed), p(out),
                           C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    return out.permute(0, 3, 1, 2).cpu()


def rnd(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("B,S,Cin,Cout", [(2, 16, 64, 64), (1, 64, 16, 128), (3, 10, 32, 16), (7, 5, 64, 64), (3, 8, 48, 192),
                                          (2, 40, 16, 64), (5, 20, 32, 1), (1, 32, 128, 256)])
def test_conv3x3_plain(B, S, Cin, Cout, prec):
    x, w, b = rnd(1, B, Cin, S, S), rnd(2, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), rnd(3, Cout)
    ref = F.conv2d(x, w, b, padding=1)
    assert rel_err(ref, run_conv(x, None, w, b, 3, 0 | prec)) < PRECS[prec]


def test_conv3x3_concat_two_sources(prec):
    x0, x1 = rnd(1, 2, 32, 40, 40), rnd(2, 2, 16, 40, 40)
    w, b = rnd(3, 128, 48, 3, 3) / 20, rnd(4, 128)
    ref = F.conv2d(torch.cat((x0, x1), 1), w, b, padding=1)
    assert rel_err(ref, run_conv(x0, x1, w, b, 3, 0 | prec)) < PRECS[prec]


def test_conv3x3_weight_standardised(prec):
    x, w, b = rnd(1, 3, 16, 10, 10), rnd(2, 16, 16, 3, 3) * 0.3 + 0.1, rnd(3, 16)
    mean = w.mean(dim=(1, 2, 3), keepdim=True)
    var = w.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
    ref = F.conv2d(x, (w - mean) * (var + 1e-5).rsqrt(), b, padding=1)
    assert rel_err(ref, run_conv(x, None, w, b, 3, 2 | prec)) < PRECS[prec]


def test_conv3x3_nearest_upsample_on_load(prec):
    x, w, b = rnd(1, 2, 32, 8, 8), rnd(2, 16, 32, 3, 3) / 17, rnd(3, 16)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1)
    assert rel_err(ref, run_conv(x, None, w, b, 3, 1 | prec, out_hw=(16, 16))) < PRECS[prec]


@pytest.mark.parametrize("S", [16, 40])
def test_pixel_unshuffle_downsample(S, prec):
    C_, Cout = 16, 32
    x, w, b = rnd(1, 2, C_, S, S), rnd(2, Cout, 4 * C_, 1, 1) / 8, rnd(3, Cout)
    y = x.reshape(2, C_, S // 2, 2, S // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(2, 4 * C_, S // 2, S // 2)
    ref = F.conv2d(y, w, b)
    assert rel_err(ref, run_conv(x, None, w, b, 1, 4 | prec, out_hw=(S // 2, S // 2))) < PRECS[prec]


def test_conv1x1_layernorm_on_load(prec):
    x, w, g = rnd(1, 2, 64, 20, 20) * 2 + 0.5, rnd(2, 384, 64, 1, 1) / 8, rnd(3, 64) * 0.2 + 1
    var = x.var(dim=1, unbiased=False, keepdim=True)
    mean = x.mean(dim=1, keepdim=True)
    xn = (x - mean) * (var + 1e-5).rsqrt() * g.view(1, -1, 1, 1)
    ref = F.conv2d(xn, w)
    assert rel_err(ref, run_conv(x, None, w, None, 1, 16 | prec, A=g)) < PRECS[prec]


def test_conv3x3_affine_silu_on_load_keeps_zero_padding(prec):
    B, Cc, S = 3, 32, 16
    x, w, b = rnd(1, B, Cc, S, S), rnd(2, 64, Cc, 3, 3) / 17, rnd(3, 64)
    A, Bv, E = rnd(4, B, Cc) * 0.5 + 1, rnd(5, B, Cc), rnd(6, B, Cc)
    t = F.silu(x * A[:, :, None, None] + Bv[:, :, None, None]) + E[:, :, None, None]
    ref = F.conv2d(t, w, b, padding=1)   # padding is applied AFTER the transform
    assert rel_err(ref, run_conv(x, None, w, b, 3, 8 | prec, A=A, Bv=Bv, E=E)) < PRECS[prec]


# ---------------------------------------------------------------- Winograd F(2x2,3x3) kernel (conv_winograd.hip)
WINO = 32 | 256          # split-bf16 x3 products on the Winograd-domain operands


@pytest.mark.parametrize("B,S,Cin,Cout", [(2, 16, 64, 64), (1, 64, 16, 128), (1, 32, 128, 256), (3, 16, 48, 192), (5, 32, 64, 64), (2, 48, 32, 68)])
def test_winograd_conv3x3_plain(B, S, Cin, Cout):
    x, w, b = rnd(1, B, Cin, S, S), rnd(2, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), rnd(3, Cout)
    ref = F.conv2d(x, w, b, padding=1)
    assert rel_err(ref, run_conv(x, None, w, b, 3, WINO)) < 1e-4


def test_winograd_asymmetric_filter_orientation():
    """One non-zero tap at a time: catches a transposed / mirrored filter or tile transform exactly."""
    x = rnd(1, 1, 16, 16, 16)
    for ky in range(3):
        for kx in range(3):
            w = torch.zeros(64, 16, 3, 3)
            w[:, :, ky, kx] = rnd(10 + ky * 3 + kx, 64, 16) / 4
            ref = F.conv2d(x, w, None, padding=1)
            assert rel_err(ref, run_conv(x, None, w, None, 3, WINO)) < 1e-4, (ky, kx)


def test_winograd_concat_standardised_upsample():
    x0, x1 = rnd(1, 2, 32, 32, 32), rnd(2, 2, 16, 32, 32)
    w, b = rnd(3, 128, 48, 3, 3) / 20, rnd(4, 128)
    ref = F.conv2d(torch.cat((x0, x1), 1), w, b, padding=1)
    assert rel_err(ref, run_conv(x0, x1, w, b, 3, WINO)) < 1e-4
    x, w, b = rnd(1, 3, 16, 16, 16), rnd(2, 64, 16, 3, 3) * 0.3 + 0.1, rnd(3, 64)
    mean, var = w.mean(dim=(1, 2, 3), keepdim=True), w.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
    ref = F.conv2d(x, (w - mean) * (var + 1e-5).rsqrt(), b, padding=1)
    assert rel_err(ref, run_conv(x, None, w, b, 3, 2 | WINO)) < 1e-4
    x, w, b = rnd(1, 2, 32, 8, 8), rnd(2, 16, 32, 3, 3) / 17, rnd(3, 16)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1)
    assert rel_err(ref, run_conv(x, None, w, b, 3, 1 | WINO, out_hw=(16, 16))) < 1e-4


@pytest.mark.parametrize("with_e", [False, True])
def test_winograd_affine_silu_on_load_keeps_zero_padding(with_e):
    B, Cc, S = 3, 32, 32
    x, w, b = rnd(1, B, Cc, S, S), rnd(2, 64, Cc, 3, 3) / 17, rnd(3, 64)
    A, Bv, E = rnd(4, B, Cc) * 0.5 + 1, rnd(5, B, Cc), (rnd(6, B, Cc) if with_e else None)
    t = F.silu(x * A[:, :, None, None] + Bv[:, :, None, None])
    if with_e:
        t = t + E[:, :, None, None]
    ref = F.conv2d(t, w, b, padding=1)   # padding is applied AFTER the transform
    assert rel_err(ref, run_conv(x, None, w, b, 3, 8 | WINO, A=A, Bv=Bv, E=E)) < 1e-4


@pytest.mark.parametrize("mode", ["plain", "affine"])
def test_winograd_large_grid_is_correct_and_bitwise_reproducible(mode):
    B, S, Cin = 32, 64, 64
    x = rnd(1, B, Cin, S, S) * 2 + 0.5
    if mode == "plain":
        w, b = rnd(2, 128, Cin, 3, 3) / 24, rnd(3, 128)
        ref = F.conv2d(x, w, b, padding=1)
        fn = lambda: run_conv(x, None, w, b, 3, WINO)
    else:
        w, b = rnd(2, 64, Cin, 3, 3) / 24, rnd(3, 64)
        A, Bv = rnd(4, B, Cin) * 0.5 + 1, rnd(5, B, Cin)
        ref = F.conv2d(F.silu(x * A[:, :, None, None] + Bv[:, :, None, None]), w, b, padding=1)
        fn = lambda: run_conv(x, None, w, b, 3, 8 | WINO, A=A, Bv=Bv)
    outs = [fn() for _ in range(3)]
    assert rel_err(ref, outs[0]) < 1e-4
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_winograd_is_refused_where_the_shape_does_not_fit():
    x, w = rnd(1, 1, 16, 40, 40), rnd(2, 64, 16, 3, 3)
    lib = _lib()
    out = torch.empty((1, 40, 40, 64), device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    xd, wd = nhwc(x).cuda(), w.cuda()
    rc = lib.hd_debug_conv(p(xd), 16, None, 0, 1, 40, 40, p(wd), None, 64, 3, WINO, None, None, None, p(out), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc != 0          # 40x40 is not made of 16x16 blocks: the engine keeps such layers on the implicit-GEMM kernel


@pytest.mark.parametrize("mode", ["plain3x3", "affine3x3", "ln1x1", "plain1x1"])
def test_large_grid_is_correct_and_bitwise_reproducible(mode, prec):
    """Thousands of co-resident workgroups (the regime of the real batch sizes): results must match
    the CPU reference AND be identical run to run (a staging race shows up only here)."""
    B, S, Cin = 32, 64, 64
    x = rnd(1, B, Cin, S, S) * 2 + 0.5
    if mode == "plain3x3":
        w, b = rnd(2, 128, Cin, 3, 3) / 24, rnd(3, 128)
        ref = F.conv2d(x, w, b, padding=1)
        fn = lambda: run_conv(x, None, w, b, 3, 0 | prec)
    elif mode == "affine3x3":
        w, b = rnd(2, 64, Cin, 3, 3) / 24, rnd(3, 64)
        A, Bv = rnd(4, B, Cin) * 0.5 + 1, rnd(5, B, Cin)
        ref = F.conv2d(F.silu(x * A[:, :, None, None] + Bv[:, :, None, None]), w, b, padding=1)
        fn = lambda: run_conv(x, None, w, b, 3, 8 | prec, A=A, Bv=Bv)
    elif mode == "ln1x1":
        w, g = rnd(2, 384, Cin, 1, 1) / 8, rnd(3, Cin) * 0.2 + 1
        var, mean = x.var(dim=1, unbiased=False, keepdim=True), x.mean(dim=1, keepdim=True)
        ref = F.conv2d((x - mean) * (var + 1e-5).rsqrt() * g.view(1, -1, 1, 1), w)
        fn = lambda: run_conv(x, None, w, None, 1, 16 | prec, A=g)
    else:
        w, b = rnd(2, 64, Cin, 1, 1) / 8, rnd(3, 64)
        ref = F.conv2d(x, w, b)
        fn = lambda: run_conv(x, None, w, b, 1, 0 | prec)
    outs = [fn() for _ in range(3)]
    assert rel_err(ref, outs[0]) < PRECS[prec]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


_BIG_REF = {}


def test_large_grid_8wave_tile_256_to_256_at_64(prec):
    """hicedrn's body layer at the grid `bench.py --workload hicedrn64` launches it on: 256 -> 256 channels at 64 x 64, 128 tiles ->
    2 048 M tiles x 2 N tiles = 4 096 workgroups of the 8-wave 256 x 128 kernel, eight 32-channel K slices per tap.  Three sampled
    tiles against torch CPU conv2d, the whole output bit for bit run to run, and a slice of the batch on its own bit for bit."""
    B, S, Cc = 128, 64, 256
    x = rnd(1, B, Cc, S, S)
    w, b = rnd(2, Cc, Cc, 3, 3) / 48, rnd(3, Cc)
    pick = [0, 77, 127]
    if "ref" not in _BIG_REF:
        _BIG_REF["ref"] = F.conv2d(x[pick], w, b, padding=1)
    out = run_conv(x, None, w, b, 3, 0 | prec)
    assert rel_err(_BIG_REF["ref"], out[pick]) < PRECS[prec]
    assert torch.equal(out, run_conv(x, None, w, b, 3, 0 | prec))
    assert torch.equal(out[40:44], run_conv(x[40:44], None, w, b, 3, 0 | prec))


@pytest.mark.parametrize("B,S,Cc", [(2, 16, 64), (3, 16, 128), (2, 16, 256), (32, 64, 64), (48, 32, 128)])
def test_linear_attention_q_side_fused(B, S, Cc):
    """softmax_d(q) -> context -> to_out -> LayerNorm -> + x as ONE convolution (context folded into the weight per
    sample, softmax in the loader, LayerNorm + residual in the epilogue) against the reference formulation
    (src/hicdiff.py:217-226, 99-108, 64-70); the large cases also have to repeat bit-for-bit."""
    lib = _lib()
    lib.hd_debug_linattn_out.restype = C.c_int
    lib.hd_debug_linattn_out.argtypes = [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p, C.c_void_p]
    heads, D = 4, 32
    q = rnd(1, B, heads * D, S, S) * 1.5
    ctx = rnd(2, B, heads, D, D) * 0.3
    wout, bias, g = rnd(3, Cc, heads * D, 1, 1) / 11, rnd(4, Cc) * 0.1, rnd(5, Cc) * 0.2 + 1
    res = rnd(6, B, Cc, S, S)
    qs = q.view(B, heads, D, S * S).softmax(dim=-2) * D ** -0.5
    o = torch.einsum("bhde,bhdn->bhen", ctx, qs).reshape(B, heads * D, S, S)
    y = F.conv2d(o, wout, bias)
    mean, var = y.mean(dim=1, keepdim=True), y.var(dim=1, unbiased=False, keepdim=True)
    ref = (y - mean) * (var + 1e-5).rsqrt() * g.view(1, -1, 1, 1) + res

    def run():
        dev = "cuda"
        t = lambda a: a.to(dev).contiguous()
        qd, cd, wd, bd, gd, rd = t(nhwc(q)), t(ctx), t(wout), t(bias), t(g), t(nhwc(res))
        out = torch.full((B, S, S, Cc), float("nan"), device=dev)
        p = lambda a: C.c_void_p(a.data_ptr())
        rc = lib.hd_debug_linattn_out(p(qd), p(cd), p(wd), p(bd), p(gd), p(rd), B, S, S, Cc, p(out),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        torch.cuda.synchronize()
        return out.permute(0, 3, 1, 2).cpu()

    outs = [run() for _ in range(3 if B >= 32 else 1)]
    assert rel_err(ref, outs[0]) < 1e-4
    for o2 in outs[1:]:
        assert torch.equal(outs[0], o2)


@pytest.mark.parametrize("tag", ["s40", "s64", "far"])
def test_tile_metrics_golden(tag):
    """hd_tile_metrics (SSIM + the moments behind mse / psnr / snr / pcc) against the values produced by the reference's
    own SSIM module and evaluation formulas (tests/golden/make_golden.py::case_metrics) and against the oracle."""
    import os
    import numpy as np
    from _util import GOLDEN
    from hicdiff_amd.Utils import metrics as M
    from hicdiff_amd.Utils.loss.SSIM import SSIM, ssim
    from oracle import metrics as OM
    g = np.load(os.path.join(GOLDEN, "metrics.npz"))
    pr, hq = torch.from_numpy(g[f"{tag}_pred"]).cuda(), torch.from_numpy(g[f"{tag}_target"]).cuda()
    m = M.batch_metrics(pr, hq)
    assert abs(m["ssim"] - float(g[f"{tag}_ssim"])) < 2e-6          # fp32 window sums in another order than conv2d
    assert abs(m["mse"] - float(g[f"{tag}_mse"])) <= 1e-6 * float(g[f"{tag}_mse"])
    assert abs(m["snr"] - float(g[f"{tag}_snr"])) <= 1e-5 * abs(float(g[f"{tag}_snr"]))
    assert abs(m["pcc"] - float(g[f"{tag}_pcc"])) < 1e-6
    assert abs(m["psnr"] - float(g[f"{tag}_psnr"])) < 1e-4
    o, h = OM.rescaled(pr.cpu()).cuda(), OM.rescaled(hq.cpu()).cuda()          # the ssim(img1, img2) entry takes [0,1] images
    each = ssim(o, h, size_average=False).cpu().numpy()
    assert np.abs(each - g[f"{tag}_ssim_each"]).max() < 2e-6
    assert abs(float(SSIM()(o, h)) - float(g[f"{tag}_ssim"])) < 2e-6
    a, b = M.tile_sums(pr, hq)[0].cpu(), M.tile_sums(pr, hq)[0].cpu()
    assert torch.equal(a, b)                                                    # deterministic reduction


def test_tile_metrics_running_log_and_errors():
    from hicdiff_amd.Utils import metrics as M
    from oracle import metrics as OM
    log = M.MetricLog()
    tot_mse, n = 0.0, 0
    for seed, B in ((1, 5), (2, 3)):
        hq = (rnd(seed, B, 1, 64, 64) * 0.4).clamp(-1, 1)
        pr = (hq + 0.1 * rnd(seed + 10, B, 1, 64, 64)).clamp(-1, 1)
        m = log.update(pr.cuda(), hq.cuda())
        ref = OM.batch_metrics(pr, hq)
        for k in ("mse", "ssim", "snr", "pcc", "psnr"):
            assert abs(m[k] - ref[k]) <= 1e-5 * max(1.0, abs(ref[k])), k
        tot_mse += ref["mse"] * B; n += B
    assert log.r["nsamples"] == 8 and abs(log.r["mse"] - tot_mse) < 1e-9
    with pytest.raises(RuntimeError):
        M.tile_sums(torch.zeros(1, 1, 8, 8), torch.zeros(1, 1, 8, 8))              # CPU tensors: no fallback
    with pytest.raises(ValueError):
        M.tile_sums(torch.zeros(1, 2, 8, 8).cuda(), torch.zeros(1, 2, 8, 8).cuda())


@pytest.mark.parametrize("B,S", [(2, 16), (3, 24), (32, 64)])
def test_linear_attention_q_chain_fused(B, S):
    """PreNorm -> to_q -> softmax_d -> context -> to_out -> LayerNorm -> + x in ONE kernel (64-channel maps) against the
    reference formulation (src/hicdiff.py:99-118, 212-226, 64-70); the large case also has to repeat bit-for-bit."""
    lib = _lib()
    lib.hd_debug_linattn_q.restype = C.c_int
    lib.hd_debug_linattn_q.argtypes = [C.c_void_p] * 7 + [C.c_int] * 3 + [C.c_void_p, C.c_void_p]
    heads, D, Cc = 4, 32, 64
    x = rnd(1, B, Cc, S, S) * 1.5 + 0.3
    ng = rnd(2, Cc) * 0.2 + 1
    wqkv = rnd(3, 3 * heads * D, Cc, 1, 1) / 6
    ctx = rnd(4, B, heads, D, D) * 0.3
    wout, bias, g = rnd(5, Cc, heads * D, 1, 1) / 11, rnd(6, Cc) * 0.1, rnd(7, Cc) * 0.2 + 1
    mean, var = x.mean(dim=1, keepdim=True), x.var(dim=1, unbiased=False, keepdim=True)
    xn = (x - mean) * (var + 1e-5).rsqrt() * ng.view(1, -1, 1, 1)
    q = F.conv2d(xn, wqkv[: heads * D])
    qs = q.view(B, heads, D, S * S).softmax(dim=-2) * D ** -0.5
    o = torch.einsum("bhde,bhdn->bhen", ctx, qs).reshape(B, heads * D, S, S)
    y = F.conv2d(o, wout, bias)
    m2, v2 = y.mean(dim=1, keepdim=True), y.var(dim=1, unbiased=False, keepdim=True)
    ref = (y - m2) * (v2 + 1e-5).rsqrt() * g.view(1, -1, 1, 1) + x

    def run():
        dev = "cuda"
        t = lambda a: a.to(dev).contiguous()
        xd, ngd, wqd, cd, wd, bd, gd = t(nhwc(x)), t(ng), t(wqkv), t(ctx), t(wout), t(bias), t(g)
        out = torch.full((B, S, S, Cc), float("nan"), device=dev)
        p = lambda a: C.c_void_p(a.data_ptr())
        rc = lib.hd_debug_linattn_q(p(xd), p(ngd), p(wqd), p(cd), p(wd), p(bd), p(gd), B, S, S, p(out),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        torch.cuda.synchronize()
        return out.permute(0, 3, 1, 2).cpu()

    outs = [run() for _ in range(3 if B >= 32 else 1)]
    assert rel_err(ref, outs[0]) < 1e-4
    for o2 in outs[1:]:
        assert torch.equal(outs[0], o2)


TILE_CASES = ["pad40", "exact64", "band8", "res10k", "small", "gap", "overlap"]


@pytest.mark.parametrize("tag", TILE_CASES)
def test_split_pieces_golden(tag, tmp_path):
    """hd_split_pieces through the reference's splitPieces signature, bit for bit against the arrays the reference's own
    function cut (tests/golden/make_golden.py::case_tiles), from a file and from a device tensor; then the stitch back."""
    import os
    import numpy as np
    from _util import GOLDEN
    from hicdiff_amd import processdata as PD
    from oracle import tiles as OT
    g = np.load(os.path.join(GOLDEN, "tiles.npz"))
    n, p, st, res = (int(v) for v in g[f"{tag}_args"])
    fn = str(tmp_path / "GSE131811_mat_full_chr_1_40000.npy")
    np.save(fn, g[f"{tag}_mat"])
    got = PD.splitPieces(fn, p, st, res)
    assert got.dtype == g[f"{tag}_tiles"].dtype and got.shape == g[f"{tag}_tiles"].shape and np.array_equal(got, g[f"{tag}_tiles"])
    dev, org = PD.split_pieces_device(torch.from_numpy(g[f"{tag}_mat"]).cuda(), p, st, res)
    assert np.array_equal(dev.cpu().numpy(), g[f"{tag}_tiles"]) and np.array_equal(org, g[f"{tag}_origins"])
    if st >= p:
        back = PD.stitch_pieces_device(dev, org, n, st).cpu().numpy()
        assert np.array_equal(back, OT.stitch_pieces(g[f"{tag}_tiles"], org, n))
        assert np.array_equal(PD.stitchPieces(got, n, p, st, res), back)
        # asymmetric tiles (a sampler's output need not be symmetric): held elements win over mirrored ones
        t2 = rnd(5, *dev.shape)
        assert np.array_equal(PD.stitch_pieces_device(t2.cuda(), org, n, st).cpu().numpy(), OT.stitch_pieces(t2.numpy(), org, n))
    else:
        with pytest.raises(ValueError):
            PD.stitch_pieces_device(dev, org, n, st)


@pytest.mark.parametrize("n,p,res", [(1000, 64, 40000), (1537, 40, 40000), (2050, 64, 10000), (777, 16, 20000)])
def test_split_stitch_round_trip_large(n, p, res):
    """Size-independent properties at chromosome scale: split == oracle; stitch(split(M)) == M inside the band and 0
    outside for a symmetric M; the result is symmetric; unaligned pitches (n % 4 != 0) take the scalar path."""
    import numpy as np
    from hicdiff_amd import processdata as PD
    from oracle import tiles as OT
    a = rnd(n, n, n).numpy()
    m = np.ascontiguousarray((a + a.T) / 2)
    dev, org = PD.split_pieces_device(torch.from_numpy(m).cuda(), p, p, res)
    assert np.array_equal(dev.cpu().numpy(), OT.split_pieces(m, p, p, res))
    back = PD.stitch_pieces_device(dev, org, n).cpu().numpy()
    assert np.array_equal(back, OT.stitch_pieces(dev.cpu().numpy(), org, n))
    band = OT.stitch_pieces(np.ones((len(org), p, p), np.float32), org, n) > 0
    assert np.array_equal(back[band], m[band]) and not back[~band].any() and np.array_equal(back, back.T)


def test_tile_module_writes_the_reference_layout(tmp_path):
    """GSE130711Module.split_numpy: Splits/GSE131811_{full,noisy,sample}_chr_<c>_<res>_piece_<S>.npy with the reference's
    shapes, the 'deno' degradation under the host generator, and the chromosome splits of the dataset."""
    import numpy as np
    from hicdiff_amd import processdata as PD
    from oracle import tiles as OT

    class Two(PD.GSE130711Module):                       # two chromosomes are enough for the layout
        chromosomes = (1, 2)
        ready_count = 1
        splits = {"all": (1, 2), "train": (1,), "val": (2,), "test": (1, 2)}

    dm = Two(batch_size=4, res=40000, piece_size=40, sigma_0=0.1, root=tmp_path)
    assert dm.dirname == f"{tmp_path}/DataFull/DataFull_Human_cell1_40000_deno_0.1"
    os.makedirs(dm.dirname + "/Full_Mats")
    mats = {}
    for c, n in ((1, 130), (2, 95)):
        a = rnd(c, n, n).numpy()
        mats[c] = np.ascontiguousarray(np.clip((a + a.T) / 4, -1, 1))
        np.save(f"{dm.dirname}/Full_Mats/GSE131811_mat_full_chr_{c}_40000.npy", mats[c])
    torch.manual_seed(11)
    dm.prepare_data()
    torch.manual_seed(11)
    for c in (1, 2):
        full = np.load(f"{dm.dirname}/Splits/GSE131811_full_chr_{c}_40000_piece_40.npy")
        noisy = np.load(f"{dm.dirname}/Splits/GSE131811_noisy_chr_{c}_40000_piece_40.npy")
        samp = np.load(f"{dm.dirname}/Splits/GSE131811_sample_chr_{c}_40000_piece_40.npy")
        ref = OT.split_pieces(mats[c], 40, 40, 40000)
        assert np.array_equal(full, ref) and noisy.shape == full.shape and samp.shape == (len(full), 1600)
        z = torch.randn(len(full), 1600)                  # the same host generator draws, in the same order
        on, os_ = OT.degrade(ref, 0.1, z.numpy())
        assert np.array_equal(noisy, on) and np.array_equal(samp, os_)
    dm.setup("test")
    lq, hq, sm, info = next(iter(dm.test_dataloader()))
    assert lq.shape == hq.shape == (4, 1, 40, 40) and sm.shape == (4, 1600) and info.tolist() == [1, 1, 1, 1]
    assert len(dm.test_set) == len(OT.tile_origins(130, 40, 40, 40000)[0]) + len(OT.tile_origins(95, 40, 40, 40000)[0])
    dm.setup(2)
    assert set(dm.test_set.info.tolist()) == {2}
    with pytest.raises(NotImplementedError):
        dm.extract_constraint_mats()
    with pytest.raises(RuntimeError):
        PD.split_pieces_device(torch.zeros(8, 8), 8, 8, 40000)   # CPU tensor: no fallback


def test_winograd_routes_whole_networks_within_the_parity_bound():
    """With the opt-in switch on, every eligible layer of the full UNet and of hicedrn (64x64 tiles) runs on the Winograd kernel
    (input modes, FiLM / residual epilogues, GroupNorm partial sums) and the epsilon forward still meets the per-forward bound."""
    from _util import golden, product_hicedrn, product_unet
    lib = _lib()
    lib.hd_debug_winograd.restype = C.c_int
    lib.hd_debug_winograd.argtypes = [C.c_int]
    g = golden("eps")
    assert lib.hd_debug_winograd(1) == 0
    try:
        for kind in ("uncond", "cond", "sr3"):
            m = product_unet(kind)
            pre = f"unet_{kind}_s64_"
            cond = g.get(pre + "cond")
            out = m(g[pre + "x"].cuda(), g[pre + "t"].cuda(), None if cond is None else cond.cuda())
            assert rel_err(g[pre + "eps"], out) < 1e-4, kind
        m = product_hicedrn("cond", 3)
        out = m(g["hicedrn_cond_n3_s64_x"].cuda(), g["hicedrn_cond_n3_s64_t"].cuda(), g["hicedrn_cond_n3_s64_cond"].cuda())
        assert rel_err(g["hicedrn_cond_n3_s64_eps"], out) < 1e-4
    finally:
        lib.hd_debug_winograd(-1)


# ---------------------------------------------------------------- two fp16 products per multiply: xh (wh + wl)

F16W2_TOL = 2e-3      # the activation is rounded once to fp16 (2^-12 relative per operand); the weight is exact to 2^-22


@pytest.mark.parametrize("ck", [32, 96], ids=["ck16", "ck32"])
@pytest.mark.parametrize("B,S,Cin,Cout", [(2, 16, 64, 64), (1, 64, 64, 64), (3, 8, 128, 256), (2, 40, 64, 128), (1, 32, 128, 256), (2, 64, 128, 64)])
def test_conv3x3_two_fp16_products(B, S, Cin, Cout, ck):
    """The 3x3 kernels' second arithmetic (conv_bf16x3_kernel.h AR = 2) over the tile variants the networks use, against torch CPU conv2d
    AND against the same convolution of the fp16-rounded activations with exact weights -- the arithmetic the mode is defined as -- at the
    split kernels' own bound."""
    x, w, b = rnd(1, B, Cin, S, S), rnd(2, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), rnd(3, Cout)
    got = run_conv(x, None, w, b, 3, 512 | ck)
    assert rel_err(F.conv2d(x, w, b, padding=1), got) < F16W2_TOL
    assert rel_err(F.conv2d(x.half().float(), w, b, padding=1), got) < 1e-4
    assert torch.equal(got, run_conv(x, None, w, b, 3, 512 | ck))


@pytest.mark.parametrize("B,S,Cin,Cout", [(2, 16, 64, 64), (1, 64, 64, 64), (3, 8, 128, 256), (1, 32, 128, 256), (2, 64, 128, 64)])
def test_conv3x3_one_fp16_product(B, S, Cin, Cout):
    """AR = 3, xh wh: against torch conv2d of the fp16-rounded activations AND fp16-rounded weights (what the mode is defined as)."""
    x, w, b = rnd(1, B, Cin, S, S), rnd(2, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), rnd(3, Cout)
    got = run_conv(x, None, w, b, 3, 1024 | 512 | 96)
    assert rel_err(F.conv2d(x.half().float(), w.half().float(), b, padding=1), got) < 1e-4
    assert rel_err(F.conv2d(x, w, b, padding=1), got) < 3e-3
    assert torch.equal(got, run_conv(x, None, w, b, 3, 1024 | 512 | 96))


@pytest.mark.parametrize("B,S,Cin,Cout,K", [(2, 16, 128, 128, 3), (1, 64, 64, 64, 3), (3, 8, 128, 256, 3), (2, 32, 128, 64, 3), (2, 40, 256, 256, 3),
                                            (2, 16, 128, 128, 1), (1, 64, 128, 64, 1), (3, 8, 256, 64, 1)])
def test_conv_one_bf16_product(B, S, Cin, Cout, K):
    """AR = 1, the training step's optional arithmetic on every 32-channel-slice tile shape (conv_bf16_plain_ck32.hip + the 8-wave form):
    against torch conv2d of the bf16-rounded activations AND bf16-rounded weights (what the mode is defined as)."""
    x, w, b = rnd(1, B, Cin, S, S), rnd(2, Cout, Cin, K, K) / (K * Cin ** 0.5), rnd(3, Cout)
    got = run_conv(x, None, w, b, K, 2048 | 96)
    r = lambda t: t.bfloat16().float()
    assert rel_err(F.conv2d(r(x), r(w), b, padding=K // 2), got) < 1e-4
    assert 1e-4 < rel_err(F.conv2d(x, w, b, padding=K // 2), got) < 2e-2          # it really is the cheaper arithmetic
    assert torch.equal(got, run_conv(x, None, w, b, K, 2048 | 96))


def test_conv_one_bf16_product_groupnorm_apply_loader():
    """... with the GroupNorm-apply + SiLU loader (+ the time-embedding term), two sources: the transformed activation is what gets rounded."""
    B, S = 2, 16
    x0, x1 = rnd(1, B, 64, S, S), rnd(2, B, 64, S, S)
    w, b = rnd(3, 128, 128, 3, 3) / (3 * 128 ** 0.5), rnd(4, 128)
    A, Bv = rnd(5, B, 128) * 0.5 + 1, rnd(6, B, 128) * 0.3
    x = torch.cat((x0, x1), 1)
    xt = F.silu(x * A[:, :, None, None] + Bv[:, :, None, None])
    got = run_conv(x0, x1, w, b, 3, 2048 | 96 | 8, A=A, Bv=Bv)
    r = lambda t: t.bfloat16().float()
    assert rel_err(F.conv2d(r(xt), r(w), b, padding=1), got) < 2e-3              # (the loader's SiLU uses the hardware exp: roundings to bf16 can flip)
    assert rel_err(F.conv2d(xt, w, b, padding=1), got) < 2e-2


def test_conv3x3_two_fp16_products_loaders_and_geometry():
    """GroupNorm-apply + SiLU loader (rounding happens AFTER the transform), concat of two sources, nearest-x2 upsample."""
    B, Cc, S = 3, 64, 32
    x, w, b = rnd(1, B, Cc, S, S), rnd(2, 64, Cc, 3, 3) / 24, rnd(3, 64)
    A, Bv = rnd(4, B, Cc) * 0.5 + 1, rnd(5, B, Cc)
    t = F.silu(x * A[:, :, None, None] + Bv[:, :, None, None])
    got = run_conv(x, None, w, b, 3, 8 | 512 | 96, A=A, Bv=Bv)
    assert rel_err(F.conv2d(t.half().float(), w, b, padding=1), got) < 1e-4
    x0, x1 = rnd(6, 2, 64, 40, 40), rnd(7, 2, 64, 40, 40)
    w2 = rnd(8, 64, 128, 3, 3) / 34
    got = run_conv(x0, x1, w2, b, 3, 512 | 96)
    assert rel_err(F.conv2d(torch.cat((x0, x1), 1).half().float(), w2, b, padding=1), got) < 1e-4
    xs = rnd(9, 2, 128, 16, 16)
    got = run_conv(xs, None, w2, b, 3, 1 | 512 | 96, out_hw=(32, 32))
    assert rel_err(F.conv2d(F.interpolate(xs, scale_factor=2, mode="nearest").half().float(), w2, b, padding=1), got) < 1e-4


def test_conv1x1_ignores_the_two_product_switch():
    """Only the 3x3 kernels have the second arithmetic: a 1x1 layer asked for it runs split-bf16 x3 (bit-identical)."""
    x, w, b = rnd(1, 2, 64, 16, 16), rnd(2, 64, 64, 1, 1) / 8, rnd(3, 64)
    assert torch.equal(run_conv(x, None, w, b, 1, 96), run_conv(x, None, w, b, 1, 96))
