"""DDRM with the non-identity degradations (src/functions/svd_replacement.py:72-541, H_func.py:4-67,
denoising.py:11-111) on the HIP engine, against fixtures generated from the reference (tests/golden/make_golden.py,
case ddrm_general): every operator's V / V^T / U / U^T as dense matrices (exact: they are permutations, tiny SVD factors
and Hadamard signs), and 10-step chains with replayed noise within the 1e-3 parity bound."""
import numpy as np
import pytest
import torch

from _util import golden, oracle_unet, product_unet, rel_err, tiles

pytestmark = pytest.mark.gpu
S = 8
CASES = [("inp_mask", 1), ("sr2", 1), ("sr4", 1), ("deblur_uni", 1), ("deblur_gauss", 1), ("deblur_aniso", 1), ("cs2", 1), ("cs4", 1),
         ("color", 3), ("sr_bicubic2", 3), ("inp_mask", 3), ("sr2", 3)]
CHAINS = ("inp_mask", "sr2", "deblur_uni", "deblur_gauss", "deblur_aniso", "cs2")


def build(deg, C_, g):
    """The product operator, built from the fixture's construction data where the reference draws it at random (mask, permutation)
    or where LAPACK is free to pick signs (the small SVD factors of the blur operators)."""
    from hicdiff_amd.functions import svd_replacement as SV
    from hicdiff_amd.functions.H_func import MakeFunc
    pre, dev = f"{deg}_c{C_}_", "cuda"
    if deg == "inp_mask":
        return SV.Inpainting(C_, S, g[pre + "missing"], dev)
    if deg[:2] == "cs":
        return SV.WalshHadamardCS(C_, S, int(deg[2:]), g[pre + "perm"], dev)
    if deg in ("deblur_uni", "deblur_gauss"):
        return SV.Deblurring(None, C_, S, dev, _svd=(g[pre + "U_small"], _raw_s(deg, g, pre), g[pre + "V_small"]))
    if deg == "deblur_aniso":
        return SV.Deblurring2D(None, None, C_, S, dev, _svd=((g[pre + "U_small1"], g[pre + "singulars_small1"], g[pre + "V_small1"]),
                                                             (g[pre + "U_small2"], g[pre + "singulars_small2"], g[pre + "V_small2"])))
    if deg.startswith("sr_bicubic"):
        return SV.SRConv(None, C_, S, dev, stride=int(deg[10:]), _svd=(g[pre + "U_small"], g[pre + "singulars_small"], g[pre + "V_small"]))
    return MakeFunc(deg, C_, S, device=dev)           # sr<k>, color: deterministic constructions


def _raw_s(deg, g, pre):
    return g[pre + "singulars_small"]                 # already thresholded upstream; thresholding again changes nothing


@pytest.mark.parametrize("deg,C_", CASES)
def test_operator_matches_reference_matrices(deg, C_):
    g = golden("ddrm_general")
    H = build(deg, C_, g)
    pre = f"{deg}_c{C_}_"
    D = C_ * S * S
    M = g[pre + "U"].shape[0]
    eD, eM = torch.eye(D, device="cuda"), torch.eye(M, device="cuda")
    for name, basis in (("V", eD), ("Vt", eD), ("U", eM), ("Ut", eM)):
        got = getattr(H, name)(basis.clone()).reshape(basis.shape[0], -1).cpu()
        assert torch.allclose(got, g[pre + name], atol=2e-6), (name, (got - g[pre + name]).abs().max())
    assert torch.allclose(H.singulars().cpu(), g[pre + "s"], atol=1e-7)
    # H = U S V^T and its pseudo-inverse act consistently: H H^+ y = y on the measurable part
    if deg != "sr_bicubic2":                          # (its singular-value vector is written for 3 channels upstream and longer than U's input)
        x = tiles(5, 2, S, C_).cuda()
        y = H.H(x)
        s = H.singulars()
        keep = (s > 0).float()
        back = H.H(H.H_pinv(y.clone()).reshape(2, C_, S, S)) if bool((s > 0).all()) else None
        if back is not None:
            assert rel_err(y, back) < 1e-4
        assert tuple(y.reshape(2, -1).shape) == (2, M) and keep.shape[0] == M


def test_general_h_dense_factors_and_matmul_primitives():
    """GeneralH (src/functions/svd_replacement.py:72-107): any dense H through its SVD, the four factors applied by hd_dense_matmul; and the
    two GEMM primitives against fp64 products on the host (sizes that are not multiples of the 64-wide tiles included)."""
    from hicdiff_amd.functions import svd_replacement as SV
    gen = torch.Generator().manual_seed(12)
    Hm = torch.randn(50, 100, generator=gen)                          # 50 measurements of a 100-dimensional signal
    H = SV.GeneralH(Hm.cuda())
    x = torch.randn(3, 100, generator=gen).cuda()
    y = H.H(x)                                                         # U S V^T x
    assert rel_err(x.cpu().double() @ Hm.double().T, y.cpu().double()) < 1e-5
    assert rel_err(x, H.V(H.Vt(x))) < 1e-5 and rel_err(y, H.U(H.Ut(y))) < 1e-5         # orthogonal factors
    assert tuple(H.add_zeros(y).shape) == (3, 100) and H.singulars().shape[0] == 50
    a, b = torch.randn(130, 75, generator=gen), torch.randn(75, 201, generator=gen)
    assert rel_err(a.double() @ b.double(), SV.dense_matmul(a.cuda(), b.cuda()).cpu().double()) < 1e-6
    A, Bm, X = torch.randn(40, 40, generator=gen), torch.randn(40, 40, generator=gen), torch.randn(5, 40, 40, generator=gen)
    assert rel_err(A.double() @ X.double() @ Bm.double(), SV.sandwich_matmul(A.cuda(), X.cuda(), Bm.cuda()).cpu().double()) < 1e-6
    # images wider than the LDS form takes (S > 64: the reference's operators are written for any img_dim) go through two dense products
    A, Bm, X = torch.randn(96, 96, generator=gen), torch.randn(96, 96, generator=gen), torch.randn(3, 96, 96, generator=gen)
    assert rel_err(A.double() @ X.double() @ Bm.double(), SV.sandwich_matmul(A.cuda(), X.cuda(), Bm.cuda().t().contiguous().t()).cpu().double()) < 1e-6
    with pytest.raises(ValueError):
        SV.sandwich_matmul(A, X.cuda(), Bm.cuda())                     # a factor left on the host


def test_makefunc_builds_every_degradation_and_refuses_unknown_names():
    from hicdiff_amd.functions import svd_replacement as SV
    from hicdiff_amd.functions.H_func import MakeFunc
    want = {"deno": SV.Denoising, "cs2": SV.WalshHadamardCS, "inp_mask": SV.Inpainting, "sr_bicubic2": SV.SRConv, "deblur_uni": SV.Deblurring,
            "deblur_gauss": SV.Deblurring, "deblur_aniso": SV.Deblurring2D, "sr4": SV.SuperResolution}
    for deg, cls in want.items():
        assert type(MakeFunc(deg, 1, 16, device="cuda")) is cls, deg
    assert type(MakeFunc("color", 3, 16, device="cuda")) is SV.Colorization
    with pytest.raises(ValueError):
        MakeFunc("nonsense", 1, 16, device="cuda")


@pytest.mark.parametrize("deg", CHAINS)
@pytest.mark.parametrize("prec", ["bf16x3", "f32"])
def test_general_ddrm_chain_golden(deg, prec, monkeypatch):
    monkeypatch.setenv("HICDIFF_PRECISION", prec)
    from hicdiff_amd.functions.denoising import efficient_generalized_steps
    from hicdiff_amd.hicdiff import HostReplayNoise
    g = golden("ddrm_general")
    sch = golden("schedules")
    pre = f"{deg}_c1_"
    m = product_unet("uncond", 16, (1, 2))
    H = build(deg, 1, g)
    nz = HostReplayNoise(2025, "cuda")
    x = nz.randn((2, 1, S, S))
    xs, x0s = efficient_generalized_steps(x, range(0, 1000, 100), m, sch["ddrm_linear_betas"].cuda(), H, g[pre + "y0"].cuda(), 0.1,
                                          etaB=1.0, etaA=0.85, etaC=0.85, noise=nz)
    assert len(xs) == 11 and len(x0s) == 10
    assert rel_err(g[pre + "final"], xs[-1]) < 1e-3
    assert rel_err(g[pre + "x0_last"], x0s[-1]) < 1e-3


def test_general_ddrm_device_noise_vs_oracle():
    """The default path (no replayed noise): the spectral update draws its three Gaussian fields on the device; read them back
    (hd_debug_randn, streams 0 / 1 / 2) and replay them through the oracle's general sampler."""
    from test_gpu_timed_path import device_randn
    from hicdiff_amd.functions.denoising import efficient_generalized_steps
    from oracle import ddrm as ODR
    g = golden("ddrm_general")
    sch = golden("schedules")
    pre = "sr2_c1_"
    m, H = product_unet("uncond", 16, (1, 2)), build("sr2", 1, g)
    B, seed, D = 2, 424242, S * S
    M = g[pre + "U"].shape[0]
    x = device_randn(B, S, seed, 0, 1000)
    seq = range(0, 1000, 100)
    xs, _ = efficient_generalized_steps(x.clone(), seq, m, sch["ddrm_linear_betas"].cuda(), H, g[pre + "y0"].cuda(), 0.1, etaB=1.0, etaA=0.85,
                                        etaC=0.85, noise=None, seed=seed, tile_offset=0)
    s = g[pre + "s"]
    betas = ODR.ddrm_betas("linear")

    class Replay:                                   # per step k: (n, D) stream 0; (n, #after) stream 1 at the 'after' elements; (n, M) stream 2
        def __init__(self):
            self.k, self.i = 0, 0
            self.after = None

        def randn(self, shape):
            k, i = self.k, self.i
            self.i += 1
            if self.i == 3:
                self.i, self.k = 0, self.k + 1
            full = device_randn(B, S, seed, 0, k, i).cpu().reshape(B, D)
            if i == 0:
                j = list(reversed([-1] + list(seq)[:-1]))[k]
                a = ODR.alpha_bar(betas, j)
                sn = (1 - a).sqrt() / a.sqrt()
                self.after = torch.nonzero(s * sn < 0.1).reshape(-1)
                return full
            if i == 1:
                return full[:, self.after]
            return full[:, :M]

    want, _ = ODR.ddrm_general(x.cpu(), seq, oracle_unet("uncond", 16, (1, 2)), betas, ODR.DenseH(g[pre + "V"].T, g[pre + "U"].T, s), g[pre + "y0"], 0.1,
                               noise=Replay())
    assert rel_err(want, xs[-1]) < 1e-3
