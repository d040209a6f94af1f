"""The path `bench.py` times and `inference.py` runs by default -- hipGraph replay of the fused step, step
scalars read from the device `StepParams` block, noise drawn on the device (Philox4x32-10 + Box-Muller) --
against the CPU oracle.

The chain's device noise is read back with the test-only `hd_debug_randn` (same key material the step
kernels use: seed, global tile index, step, noise stream) and replayed through the oracle in the reference's
call order (src/hicdiff.py:599,607; src/functions/denoising.py:92,96,100), so the comparison covers the
graph capture, the replayed coefficients and the in-kernel generator at once.

Tolerance (BASELINE.json north_star): 1e-3 relative = max|got - ref| / max|ref| on chains.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from _util import diffusion_class, oracle_hicedrn, oracle_unet, product_hicedrn, product_unet, rel_err, tiles

pytestmark = pytest.mark.gpu
CHAIN_TOL = 1e-3


def _lib():
    from hicdiff_amd import _lib as L
    lib = L.load()
    lib.hd_debug_randn.restype = C.c_int
    lib.hd_debug_randn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
    return lib


def device_randn(B, S, seed, tile_offset, step, nstream=0):
    out = torch.empty((B, 1, S, S), device="cuda", dtype=torch.float32)
    rc = _lib().hd_debug_randn(C.c_void_p(out.data_ptr()), B, S, seed, tile_offset, step, nstream,
                               C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    return out


class AncestralDeviceNoise:
    """oracle-side noise object: the k-th `randn` call returns the device draw the fused chain uses at that
    point -- x_T is keyed by step = T, the z of reverse step t by step = t (hicdiff_amd/_diffusion.py)."""

    def __init__(self, B, S, T, seed, tile_offset=0):
        self.B, self.S, self.seed, self.off = B, S, seed, tile_offset
        self.steps = iter([T] + list(range(T - 1, 0, -1)))

    def randn(self, shape):
        assert tuple(shape) == (self.B, 1, self.S, self.S)
        return device_randn(self.B, self.S, self.seed, self.off, next(self.steps)).cpu()


class DdrmDeviceNoise:
    """three draws per executed step k (0-based), noise streams 0 / 1 / 2 = 'missing' / 'after' / 'before'."""

    def __init__(self, B, S, seed):
        self.B, self.S, self.seed, self.k, self.i = B, S, seed, 0, 0

    def randn(self, shape):
        n, d = shape
        stream = self.i
        self.i += 1
        k = self.k
        if self.i == 3:
            self.i, self.k = 0, self.k + 1
        if d == 0:
            return torch.empty((n, 0))
        return device_randn(self.B, self.S, self.seed, 0, k, stream).cpu().reshape(n, d)


_ORACLE = {}


def oracle_once(key, fn):
    """The CPU oracle's side of a test does not depend on the arithmetic under test: compute it once per session."""
    if key not in _ORACLE:
        _ORACLE[key] = fn()
    return _ORACLE[key]


@pytest.fixture(params=["bf16x3", "f32"])
def precision(request, monkeypatch):
    monkeypatch.setenv("HICDIFF_PRECISION", request.param)
    return request.param


@pytest.fixture(autouse=True)
def _graphs_forced_on(monkeypatch):
    """These tests are about the hipGraph replay of the fused steps.  Contexts replay by default only from 150 k pixels per step on (smaller
    steps run faster eagerly: engine.hip run_step); the environment switch, read at hd_create, forces the replay for the small shapes used here."""
    monkeypatch.setenv("HICDIFF_GRAPHS", "1")


def _set_graphs(model, on):
    eng = model.engine(torch.device("cuda", torch.cuda.current_device()))
    assert eng.lib.hd_set_graphs(eng.ctx, 1 if on else 0) == 0


# ---------------------------------------------------------------- graph-replayed chains vs oracle

def test_graph_replayed_ancestral_chain_uncond_vs_oracle(precision):
    B, S, T, seed = 2, 40, 50, 4242
    net = product_unet("uncond")
    d = diffusion_class("uncond")(net, image_size=S, timesteps=T, loss_type="l2", beta_schedule="linear").cuda()
    d.seed = seed
    assert d.noise_source is None                       # -> device Philox, hipGraph replay from the third step on
    got = d.sample(torch.zeros(B, 1, S, S))
    from oracle import diffusion as OD
    ref = OD.DiffusionRef(oracle_unet("uncond"), image_size=S, timesteps=T, beta_schedule="linear", loss_type="l2")
    want = oracle_once("uncond50", lambda: ref.p_sample_loop((B, 1, S, S), AncestralDeviceNoise(B, S, T, seed)))
    assert rel_err(want, got) < CHAIN_TOL
    # the same chain launched eagerly (no graph) is bit-identical: capture / replay changes nothing
    _set_graphs(net, False)
    eager = d.sample(torch.zeros(B, 1, S, S))
    _set_graphs(net, True)
    assert torch.equal(eager, got)
    # and a second graph run repeats (the StepParams block is rewritten every step, nothing stale survives a chain)
    assert torch.equal(d.sample(torch.zeros(B, 1, S, S)), got)


def test_graph_replayed_ancestral_chain_cond_vs_oracle(precision):
    B, S, T, seed = 2, 40, 50, 99
    net = product_unet("cond")
    d = diffusion_class("cond")(net, image_size=S, timesteps=T, loss_type="l2", beta_schedule="linear").cuda()
    d.seed = seed
    lq = tiles(17, B, S)
    got = d.super_resolution(lq.cuda())
    from oracle import diffusion as OD
    ref = OD.DiffusionRef(oracle_unet("cond"), image_size=S, timesteps=T, beta_schedule="linear", loss_type="l2", kind="cond")
    want = oracle_once("cond50", lambda: ref.p_sample_loop(lq, AncestralDeviceNoise(B, S, T, seed)))
    assert rel_err(want, got) < CHAIN_TOL
    _set_graphs(net, False)
    eager = d.super_resolution(lq.cuda())
    _set_graphs(net, True)
    assert torch.equal(eager, got)


def test_graph_replayed_chain_with_tile_offset_vs_oracle(precision):
    """A shard's chain (tile_offset != 0) draws the noise of its GLOBAL tiles -- also on the replayed path."""
    B, S, T, seed, off = 3, 16, 50, 7, 5
    net = product_unet("uncond", 16, (1, 2))
    d = diffusion_class("uncond")(net, image_size=S, timesteps=T, loss_type="l2", beta_schedule="linear").cuda()
    d.seed, d.tile_offset = seed, off
    got = d.sample(torch.zeros(B, 1, S, S))
    from oracle import diffusion as OD
    ref = OD.DiffusionRef(oracle_unet("uncond", 16, (1, 2)), image_size=S, timesteps=T, beta_schedule="linear", loss_type="l2")
    want = oracle_once("offset50", lambda: ref.p_sample_loop((B, 1, S, S), AncestralDeviceNoise(B, S, T, seed, off)))
    assert rel_err(want, got) < CHAIN_TOL


@pytest.mark.parametrize("net_kind", ["unet", "hicedrn3"])
@pytest.mark.parametrize("sigma_0", [0.1, 1.0])
def test_graph_replayed_ddrm_chain_vs_oracle(net_kind, sigma_0, precision):
    """inference.py -u 1 as it runs by default: DDRM 'deno', 50 of 1000 steps, device noise, graph replay.
    sigma_0 = 0.1 crosses from the 'before' to the 'after' branch inside the chain, 1.0 stays 'after' longer."""
    from hicdiff_amd.functions.H_func import MakeFunc
    from hicdiff_amd.functions.denoising import efficient_generalized_steps
    from oracle import ddrm as ODD
    B, S, seed = 2, 40, 31337
    m = product_unet("uncond") if net_kind == "unet" else product_hicedrn("uncond", 3)
    ref_model = oracle_unet("uncond") if net_kind == "unet" else oracle_hicedrn("uncond", 3)
    betas = ODD.ddrm_betas("linear", 1000)
    hq = tiles(5, B, S)
    y0 = (hq + sigma_0 * torch.randn(hq.shape, generator=torch.Generator().manual_seed(6))).clamp(-1, 1)
    x = device_randn(B, S, seed, 0, 1000)          # any start state; the same tensor goes to both sides
    H = MakeFunc("deno", 1, S, device="cuda")
    seq = range(0, 1000, 20)
    xs, x0s = efficient_generalized_steps(x.clone(), seq, m, betas.cuda(), H, y0.cuda(), sigma_0, etaB=1.0, etaA=0.85, etaC=0.85,
                                          noise=None, seed=seed)
    want, want_x0 = oracle_once(("ddrm", net_kind, sigma_0), lambda: ODD.ddrm_denoise(x.cpu(), list(seq), ref_model, betas, y0, sigma_0,
                                                                                     noise=DdrmDeviceNoise(B, S, seed)))
    assert rel_err(want, xs[-1]) < CHAIN_TOL
    assert rel_err(want_x0, x0s[-1]) < CHAIN_TOL
    _set_graphs(m, False)
    xs2, _ = efficient_generalized_steps(x.clone(), seq, m, betas.cuda(), H, y0.cuda(), sigma_0, etaB=1.0, etaA=0.85, etaC=0.85,
                                         noise=None, seed=seed)
    _set_graphs(m, True)
    assert torch.equal(xs2[-1], xs[-1])


@pytest.mark.parametrize("net_kind", ["unet", "hicedrn3"])
@pytest.mark.parametrize("sigma_0", [0.1, 1.0])
def test_ddrm_skip_inert_steps_is_bit_identical(net_kind, sigma_0):
    """--skip-inert-steps (opt-in): with etaB = 1, while sigma_next > sigma_0 the update is y + sqrt(sigma_next^2 - sigma_0^2) z whatever the
    network says (src/functions/denoising.py:99-100); those steps run the update kernel alone.  The final tiles and the final x0 estimate
    equal the full run's bit for bit, with device noise and with replayed noise; 47 (sigma_0 = 0.1) / 36 (1.0) of the 50 steps are skipped."""
    from hicdiff_amd import _lib as L
    from hicdiff_amd.functions.H_func import MakeFunc
    from hicdiff_amd.functions.denoising import efficient_generalized_steps
    from oracle import ddrm as ODD
    B, S, seed = 2, 40, 31337
    m = product_unet("uncond") if net_kind == "unet" else product_hicedrn("uncond", 3)
    betas = ODD.ddrm_betas("linear", 1000).cuda()
    hq = tiles(5, B, S)
    y0 = (hq + sigma_0 * torch.randn(hq.shape, generator=torch.Generator().manual_seed(6))).clamp(-1, 1).cuda()
    x = device_randn(B, S, seed, 0, 1000)
    H = MakeFunc("deno", 1, S, device="cuda")
    seq = range(0, 1000, 20)
    ab = (1 - torch.cat([torch.zeros(1), betas.cpu()])).cumprod(0)
    inert = sum(1 for j in [-1] + list(seq)[:-1] if float(((1 - ab[j + 1]) / ab[j + 1]).sqrt()) > sigma_0)
    assert inert == (47 if sigma_0 == 0.1 else 36)

    def run(skip, noise=None):
        xs, x0s = efficient_generalized_steps(x.clone(), seq, m, betas, H, y0, sigma_0, etaB=1.0, etaA=0.85, etaC=0.85, noise=noise, seed=seed,
                                              keep="last", skip_inert_steps=skip)
        return xs[-1].clone(), x0s[-1].clone()

    full, full0 = run(False)
    fast, fast0 = run(True)
    assert torch.equal(full, fast) and torch.equal(full0, fast0)
    from hicdiff_amd.hicdiff import HostReplayNoise
    full, full0 = run(False, HostReplayNoise(3, "cuda"))
    fast, fast0 = run(True, HostReplayNoise(3, "cuda"))
    assert torch.equal(full, fast) and torch.equal(full0, fast0)
    with pytest.raises(ValueError):
        efficient_generalized_steps(x.clone(), seq, m, betas, H, y0, sigma_0, etaB=1.0, etaA=0.85, etaC=0.85, keep="all", skip_inert_steps=True)
    # the library refuses the switch on a step whose update reads the network
    eng = _eng(m)
    co = L.HdDdrmCoef()
    co.sqrt_at, co.sqrt_1m_at, co.sqrt_at_next, co.sigma_next, co.sigma_0 = 0.8, 0.6, 0.85, 0.62, 1.0      # sigma_next < sigma_0
    co.etaA, co.etaB, co.etaC, co.time_value, co.skip_network = 0.85, 1.0, 0.85, 300.0, 1
    with pytest.raises(L.HdError):
        eng.ddrm_step(x.clone(), y0, None, co, None, seed=1, tile_offset=0, step=0)


@pytest.mark.parametrize("eta", [0.0, 0.5])
def test_graph_replayed_ddim_vs_oracle(eta):
    """DDIM (src/hicdiff.py:622-664) on the fused step: 20 of 1000 steps with device noise, graph replay, against the oracle over the
    same noise (x_T keyed by step = T, the z of a step by its timestep), and bit-identical to the eager launch."""
    from oracle import diffusion as OD
    B, S, T, n, seed = 2, 40, 1000, 20, 606
    net = product_unet("uncond")
    d = diffusion_class("uncond")(net, image_size=S, timesteps=T, sampling_timesteps=n, loss_type="l2", beta_schedule="sigmoid",
                                  ddim_sampling_eta=eta).cuda()
    d.seed = seed
    got = d.sample(torch.zeros(B, 1, S, S))
    times = list(reversed(torch.linspace(-1, T - 1, steps=n + 1).int().tolist()))

    class Replay:
        def __init__(self):
            self.keys = iter([T] + [t for t, tn in zip(times[:-1], times[1:]) if tn >= 0])

        def randn(self, shape):
            return device_randn(B, S, seed, 0, next(self.keys)).cpu()

    ref = OD.DiffusionRef(oracle_unet("uncond"), image_size=S, timesteps=T, beta_schedule="sigmoid", sampling_timesteps=n, ddim_sampling_eta=eta)
    want = ref.ddim_sample((B, 1, S, S), Replay())
    assert rel_err(want, got) < CHAIN_TOL
    _set_graphs(net, False)
    eager = d.sample(torch.zeros(B, 1, S, S))
    _set_graphs(net, True)
    assert torch.equal(eager, got)


def test_precision_switch_drops_captured_graphs():
    """hd_set_precision between two chains on the same tensors: the second chain must run the new arithmetic
    (graphs are keyed by tensor addresses, which stay the same here).  DDRM steps: no clamp hides the difference."""
    from hicdiff_amd import _lib as L
    B, S = 2, 16
    net = product_unet("uncond", 16, (1, 2))
    eng = net.engine(torch.device("cuda", torch.cuda.current_device()))
    eng.set_precision(L.HD_PRECISION_BF16X3)
    keep = device_randn(B, S, 3, 0, 1)
    y = (tiles(4, B, S).cuda() * 0.5).contiguous()
    x, x0 = keep.clone(), torch.empty_like(keep)
    co = L.HdDdrmCoef()
    co.sqrt_at, co.sqrt_1m_at, co.sqrt_at_next, co.sigma_next, co.sigma_0 = 0.8, 0.6, 0.85, 0.62, 1.0   # sigma_next < sigma_0: the branch that keeps x0_t
    co.etaA, co.etaB, co.etaC, co.time_value = 0.85, 1.0, 0.85, 300.0

    def chain():
        x.copy_(keep)
        for k in range(6):                  # call 1 eager, call 2 captures, calls 3.. replay
            eng.ddrm_step(x, y, None, co, x0, seed=11, tile_offset=0, step=k)
        return x.clone(), x0.clone()

    fast, fast0 = chain()
    eng.set_precision(L.HD_PRECISION_F32)
    exact, exact0 = chain()                 # same tensor addresses: a stale graph would replay the bf16x3 kernels
    _set_graphs(net, False)
    exact_eager, _ = chain()
    _set_graphs(net, True)
    eng.set_precision(L.HD_PRECISION_BF16X3)
    again, _ = chain()
    assert torch.equal(exact, exact_eager)
    assert not torch.equal(exact0, fast0)
    assert torch.equal(again, fast)
    assert rel_err(exact, fast) < CHAIN_TOL


# ---------------------------------------------------------------- two half-batch chains on two streams (hd_chain_begin / hd_chain_end)

def _eng(model):
    return model.engine(torch.device("cuda", torch.cuda.current_device()))


@pytest.mark.parametrize("kind", ["uncond", "cond", "sr3"])
def test_two_half_batch_chains_equal_the_single_chain(kind, precision):
    """A replayed chain cut into two half-batch chains that advance side by side on two streams of the context (engine.hip lanes) is the
    single chain bit for bit: 6 tiles -> 4 + 2, noise keyed by the global tile, tile_offset != 0.  The single chain is the one the oracle
    tests above pin."""
    B, S, T, seed = 6, 40, 30, 515
    net = product_unet(kind, 32, (1, 2, 4))
    d = diffusion_class(kind)(net, image_size=S, timesteps=T, loss_type="l2", beta_schedule="linear").cuda()
    d.seed, d.tile_offset = seed, 3
    eng = _eng(net)
    lq = tiles(21, B, S).cuda()
    run = (lambda: d.sample(torch.zeros(B, 1, S, S))) if kind == "uncond" else (lambda: d.super_resolution(lq))
    eng.set_chains(1)
    assert eng.chains_for(B, S) == 1
    one = run()
    eng.set_chains(2)
    assert eng.chains_for(B, S) == 2
    two = run()
    assert torch.equal(one, two)
    assert torch.equal(two, run())                      # and repeats
    eng.set_chains(3)                                   # 2 + 2 + 2
    assert eng.chains_for(B, S) == 3
    assert torch.equal(one, run())
    # the per-step form without a bracket (p_sample: fork and join around every step) goes through the same lanes
    x = torch.randn(B, 1, S, S, generator=torch.Generator().manual_seed(1)).cuda()
    cond = lq if kind != "uncond" else None
    outs = []
    for n in (1, 2):
        eng.set_chains(n)
        outs.append([d.p_sample(x, 7, cond), d.p_sample(x, 7, cond), d.p_sample(x, 7, cond)][-1])     # eager, capture, replay
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_two_half_batch_ddrm_chains_equal_the_single_chain(precision):
    from hicdiff_amd.functions.H_func import MakeFunc
    from hicdiff_amd.functions.denoising import efficient_generalized_steps
    from oracle import ddrm as ODD
    B, S, seed, sigma_0 = 6, 40, 77, 0.1
    m = product_hicedrn("uncond", 2)
    betas = ODD.ddrm_betas("linear", 1000).cuda()
    y0 = (tiles(5, B, S) + sigma_0 * torch.randn((B, 1, S, S), generator=torch.Generator().manual_seed(6))).clamp(-1, 1).cuda()
    x = device_randn(B, S, seed, 0, 1000)
    H = MakeFunc("deno", 1, S, device="cuda")
    eng = _eng(m)
    outs = []
    for n in (1, 2, 2):
        eng.set_chains(n)
        xs, x0s = efficient_generalized_steps(x.clone(), range(0, 1000, 50), m, betas, H, y0, sigma_0, etaB=1.0, etaA=0.85, etaC=0.85,
                                              noise=None, seed=seed, keep="last")
        outs.append((xs[-1].clone(), x0s[-1].clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[1][0], outs[2][0])


def test_chain_bracket_mixes_with_steps_on_the_callers_stream():
    """Inside a bracket a step that must run on the caller's stream (replayed noise) joins the lanes first and the next replayed step
    forks again: the mixed sequence equals the same sequence issued step by step without a bracket."""
    import contextlib
    from hicdiff_amd import _lib as L
    B, S = 6, 16
    net = product_unet("uncond", 16, (1, 2))
    eng = _eng(net)
    eng.set_chains(2)
    start = device_randn(B, S, 3, 0, 1)
    z = device_randn(B, S, 4, 0, 2)
    co = L.HdDdpmCoef()
    co.sqrt_recip_alphas_cumprod, co.sqrt_recipm1_alphas_cumprod = 1.2, 0.66
    co.posterior_mean_coef1, co.posterior_mean_coef2, co.sigma, co.time_value = 0.3, 0.69, 0.2, 400.0

    def sequence(bracket):
        x = start.clone()
        ctx = eng.chain(B, S) if bracket else contextlib.nullcontext()
        with ctx:
            for k in range(8):
                eng.ddpm_step(x, None, z if k in (3, 6) else None, co, None, seed=9, tile_offset=0, step=k + 1)
        return x.clone()

    plain = sequence(False)
    assert torch.equal(sequence(True), plain)
    assert torch.equal(sequence(True), plain)


def test_chain_bracket_state_errors():
    from hicdiff_amd import _lib as L
    net = product_unet("uncond", 16, (1, 2))
    eng = _eng(net)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert eng.lib.hd_chain_end(eng.ctx, st) == L.HD_ESTATE
    assert eng.lib.hd_chain_begin(eng.ctx, st) == 0
    assert eng.lib.hd_chain_begin(eng.ctx, st) == L.HD_ESTATE
    assert eng.lib.hd_reserve(eng.ctx, 8, 16) == L.HD_ESTATE
    assert eng.lib.hd_set_chains(eng.ctx, 2) == L.HD_ESTATE
    assert b"bracket" in eng.lib.hd_last_error(eng.ctx)
    assert eng.lib.hd_chain_end(eng.ctx, st) == 0
    assert eng.lib.hd_set_chains(eng.ctx, 5) == L.HD_EINVAL


@pytest.mark.parametrize("kind", ["uncond", "cond"])
def test_two_chains_at_bench_batch_size_equal_the_single_chain(kind):
    """bench.py's default workload as it now runs: 256 tiles of 64x64 as two 128-tile chains (the library's default rule: every replayed
    step), six steps inside a bracket, against the single 256-tile chain."""
    B, S, T = 256, 64, 1000
    net = product_unet(kind)
    d = diffusion_class(kind)(net, image_size=S, timesteps=T, loss_type="l2", beta_schedule="linear").cuda()
    eng = _eng(net)
    assert eng.chains_for(B, S) == 2                    # the default
    cond = tiles(33, B, S).cuda() if kind == "cond" else None
    start = device_randn(B, S, 1234, 0, T)

    def six_steps():
        x = start.clone()
        with eng.chain(B, S):
            for t in range(T - 1, T - 7, -1):
                d._step_inplace(x, t, cond, eng=eng)
        return x.clone()

    two = six_steps()
    eng.set_chains(1)
    one = six_steps()
    assert torch.equal(one, two)


# ---------------------------------------------------------------- the device generator itself

def test_device_gaussian_moments_and_ks():
    """10^7 draws of the Philox4x32-10 + Box-Muller generator (fast-math log / sincos): moments and a
    Kolmogorov-Smirnov bound against N(0,1).  Standard errors at n = 1e7: mean 3.2e-4, variance 4.5e-4,
    skewness 7.7e-4, excess kurtosis 1.5e-3; bounds are 5 sigma."""
    from scipy import stats
    n_tiles, S = 2442, 64                       # 2442 * 4096 = 10 002 432 draws
    z = device_randn(n_tiles, S, 1234, 0, 17).double().flatten()
    n = z.numel()
    mean, var = z.mean().item(), z.var().item()
    zc = (z - mean) / var ** 0.5
    skew, kurt = (zc ** 3).mean().item(), (zc ** 4).mean().item() - 3.0
    assert abs(mean) < 5 * n ** -0.5, mean
    assert abs(var - 1) < 5 * (2 / n) ** 0.5, var
    assert abs(skew) < 5 * (6 / n) ** 0.5, skew
    assert abs(kurt) < 5 * (24 / n) ** 0.5, kurt
    zs = z.cpu().numpy()
    ks = stats.kstest(zs, "norm")
    assert ks.statistic < 1.63 / n ** 0.5, ks              # 1 % critical value of the KS statistic
    # tails: P(|z| > 4) = 6.33e-5 -> 633 expected in 1e7, sd 25; the largest of 1e7 normals sits near 5.3
    tail = int((np.abs(zs) > 4).sum())
    assert abs(tail - 6.334e-5 * n) < 6 * (6.334e-5 * n) ** 0.5, tail
    assert 4.6 < np.abs(zs).max() < 6.5, np.abs(zs).max()
    # Box-Muller pairs (cos, sin of one angle / two radii per counter) are uncorrelated
    q = z.reshape(-1, 4)
    for a in range(4):
        for b in range(a + 1, 4):
            r = (q[:, a] * q[:, b]).mean().item()
            assert abs(r) < 5 * (n / 4) ** -0.5, (a, b, r)


def test_device_gaussian_streams_are_distinct():
    """Different (tile, step, noise stream, seed) give different, uncorrelated fields; the same key repeats."""
    S = 64
    base = device_randn(4, S, 5, 0, 9, 0)
    assert torch.equal(base, device_randn(4, S, 5, 0, 9, 0))
    others = {
        "tile": device_randn(4, S, 5, 4, 9, 0), "step": device_randn(4, S, 5, 0, 10, 0), "stream1": device_randn(4, S, 5, 0, 9, 1),
        "stream2": device_randn(4, S, 5, 0, 9, 2), "seed": device_randn(4, S, 6, 0, 9, 0), "seed_hi": device_randn(4, S, 5 + (1 << 32), 0, 9, 0),
    }
    n = base.numel()
    for name, o in others.items():
        assert not torch.equal(o, base), name
        r = (o.double() * base.double()).mean().item()
        assert abs(r) < 5 * n ** -0.5, (name, r)
    # tile_offset is a plain shift of the global tile index
    assert torch.equal(device_randn(4, S, 5, 2, 9, 0)[:2], base[2:])
    # tiles inside one call differ from each other
    assert not torch.equal(base[0], base[1])


# ---------------------------------------------------------------- BASELINE batch size: B = 256, S = 64

@pytest.mark.parametrize("kind", ["uncond", "cond", "sr3"])
def test_eps_at_bench_batch_size_vs_oracle(kind, precision):
    """BASELINE configs[1..3]'s batch: one 256-tile, 64x64 forward; four sampled tiles of it against the
    oracle (per-forward bound of tests/test_gpu_parity.py) and slices of the batch bit for bit.  SR3 (configs[2]): the time input is
    the continuous noise level, one fp32 per tile (src/hicdiff_sr3.py:634-637)."""
    B, S = 256, 64
    m, ref = product_unet(kind), oracle_unet(kind)
    x = tiles(256, B, S)
    cond = tiles(257, B, S) if kind != "uncond" else None
    t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(8))
    if kind == "sr3":
        t = torch.rand((B, 1), generator=torch.Generator().manual_seed(8)) * 0.98 + 0.01
    xd, td, cd = x.cuda(), t.cuda(), None if cond is None else cond.cuda()
    full = m(xd, td, cd)
    pick = torch.tensor([0, 101, 200, 255])
    want = oracle_once(("b256", kind), lambda: ref(x[pick], t[pick], None if cond is None else cond[pick]))
    assert rel_err(want, full[pick.cuda()]) < 1e-4
    assert torch.equal(full, m(xd, td, cd))
    part = m(xd[96:160], td[96:160], None if cd is None else cd[96:160])
    assert torch.equal(full[96:160], part)


@pytest.mark.parametrize("kind", ["uncond", "cond", "sr3"])
def test_hicedrn_eps_at_bench_batch_size_vs_oracle(kind, precision):
    """BASELINE configs[3]'s workload at its real grid: hicedrn on 256 tiles of 64x64 -- 4 096 M tiles x 2 N tiles of the 8-wave 256 x 128
    kernel with eight K slices per tap, the grid `bench.py --workload hicedrn64` launches 65 times per step.  Three blocks keep the CPU
    oracle in seconds; four sampled tiles against it, the whole batch bit for bit run to run, and a slice of the batch on its own."""
    B, S = 256, 64
    m, ref = product_hicedrn(kind, 3), oracle_hicedrn(kind, 3)
    x = tiles(258, B, S)
    cond = tiles(259, B, S) if kind != "uncond" else None
    t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(9))
    if kind == "sr3":
        t = torch.rand((B, 1), generator=torch.Generator().manual_seed(9)) * 0.98 + 0.01
    xd, td, cd = x.cuda(), t.cuda(), None if cond is None else cond.cuda()
    full = m(xd, td, cd)
    pick = torch.tensor([0, 101, 200, 255])
    want = oracle_once(("hicedrn3_b256", kind), lambda: ref(x[pick], t[pick], None if cond is None else cond[pick]))
    assert rel_err(want, full[pick.cuda()]) < 1e-4
    assert torch.equal(full, m(xd, td, cd))
    part = m(xd[96:160], td[96:160], None if cd is None else cd[96:160])
    assert torch.equal(full[96:160], part)


def test_hicedrn32_at_bench_batch_size_properties_and_two_tiles_vs_oracle(precision):
    """The network `bench.py --workload hicedrn64` times, all 32 blocks, on its 256 tiles: two sampled tiles against the oracle, the batch
    bit for bit run to run, a slice on its own bit for bit."""
    B, S = 256, 64
    m, ref = product_hicedrn("uncond", 32), oracle_hicedrn("uncond", 32)
    x = tiles(260, B, S)
    t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(10))
    xd, td = x.cuda(), t.cuda()
    full = m(xd, td)
    assert bool(torch.isfinite(full).all())
    pick = torch.tensor([7, 250])
    want = oracle_once("hicedrn32_b256", lambda: ref(x[pick], t[pick], None))
    assert rel_err(want, full[pick.cuda()]) < 1e-4
    assert torch.equal(full, m(xd, td))
    assert torch.equal(full[96:160], m(xd[96:160], td[96:160]))


def test_hicedrn32_eps_at_64_vs_oracle(precision):
    """BASELINE configs[3]'s network at its tile size: all 32 blocks at 64x64 (the golden fixture pins 32 blocks at
    40x40 and 3 blocks at 64x64)."""
    m, ref = product_hicedrn("uncond", 32), oracle_hicedrn("uncond", 32)
    x = tiles(11, 2, 64)
    t = torch.tensor([3, 950])
    assert rel_err(oracle_once("hicedrn32_64", lambda: ref(x, t, None)), m(x.cuda(), t.cuda())) < 1e-4


# ---------------------------------------------------------------- full-length chain: drift over 1000 steps

_DRIFT = {}


def _oracle_1000(B, S, T, seed):
    key = (B, S, T, seed)
    if key not in _DRIFT:
        from oracle import diffusion as OD
        ref = OD.DiffusionRef(oracle_unet("uncond"), image_size=S, timesteps=T, beta_schedule="linear", loss_type="l2")
        _DRIFT[key] = ref.p_sample_loop((B, 1, S, S), AncestralDeviceNoise(B, S, T, seed), keep_every=100)
    return _DRIFT[key]


def test_full_length_chain_drift_vs_oracle(precision, capsys):
    """BASELINE's headline is a 1000-step chain: the full-size UNet, 2 tiles of 40x40, T = 1000 with device noise on
    the graph-replayed path against the CPU oracle over the same noise.  Prints the drift every 100 steps (DESIGN.md
    section 2 quotes it)."""
    B, S, T, seed = 2, 40, 1000, 2026
    net = product_unet("uncond")
    d = diffusion_class("uncond")(net, image_size=S, timesteps=T, loss_type="l2", beta_schedule="linear").cuda()
    d.seed = seed
    stack = d.sample(torch.zeros(B, 1, S, S), return_all_timesteps=True)      # (B, T+1, 1, S, S); index T - t = state after step t
    want, kept = _oracle_1000(B, S, T, seed)
    drift = {t: rel_err(kept[t], stack[:, T - t]) for t in sorted(kept, reverse=True)}
    with capsys.disabled():
        print(f"\n[drift {precision}] " + " ".join(f"t={t}:{e:.2e}" for t, e in drift.items()))
    assert rel_err(want, stack[:, T]) < CHAIN_TOL
    assert max(drift.values()) < CHAIN_TOL


def test_early_band_schedule_is_what_the_long_chain_runs_and_nothing_else():
    """The precision schedule (hicdiff_amd/_diffusion.py:_early_band): in the 3x3 convolutions one fp16 product for t >= 3T / 4 and two for
    T / 2 <= t < 3T / 4 of chains of at least 500 steps; 50-step chains, the second half, a network that opts out and a switched-off schedule keep three products."""
    from hicdiff_amd import _lib as L
    d = diffusion_class("uncond")(product_unet("uncond", 16, (1, 2)), image_size=16, timesteps=1000, loss_type="l2", beta_schedule="linear").cuda()
    assert [d._coef(t).arith for t in (999, 750, 749, 500, 499, 0)] == [L.HD_ARITH_F16W1, L.HD_ARITH_F16W1, L.HD_ARITH_F16W2, L.HD_ARITH_F16W2,
                                                                       L.HD_ARITH_F16W2_LOW, L.HD_ARITH_F16W2_LOW]
    d.late_band_low_f16 = False
    assert d._coef(499).arith == L.HD_ARITH_DEFAULT and d._coef(999).arith == L.HD_ARITH_F16W1
    d.late_band_low_f16 = True
    d.early_band_f16 = False
    assert d._coef(999).arith == L.HD_ARITH_DEFAULT and d._coef(10).arith == L.HD_ARITH_DEFAULT
    d50 = diffusion_class("uncond")(product_unet("uncond", 16, (1, 2)), image_size=16, timesteps=50, loss_type="l2", beta_schedule="linear").cuda()
    assert d50._coef(49).arith == L.HD_ARITH_DEFAULT
    dh = diffusion_class("uncond")(product_hicedrn("uncond", 2), image_size=16, timesteps=1000, loss_type="l2", beta_schedule="linear").cuda()
    assert dh._coef(999).arith == L.HD_ARITH_F16W1 and dh._coef(600).arith == L.HD_ARITH_F16W2 and dh._coef(10).arith == L.HD_ARITH_F16W2   # hicedrn: every step of a long chain
    dh.model.EARLY_BAND_OK = False                      # a network opts out by its class attribute
    assert dh._coef(999).arith == L.HD_ARITH_DEFAULT
    # one step each way on the same state: the early-band step differs from the three-product step (the switch is live) within its own bound
    x = device_randn(4, 16, 1, 0, 1000)
    d.early_band_f16 = True
    a, _ = d.p_sample(x, 900)                          # one product
    a2, _ = d.p_sample(x, 600)                         # two
    d.early_band_f16 = False
    b, _ = d.p_sample(x, 900)
    b2, _ = d.p_sample(x, 600)
    assert not torch.equal(a, b) and rel_err(b, a) < 1e-3 and not torch.equal(a2, b2) and rel_err(b2, a2) < 1e-3
    # the schedule was measured on the LINEAR beta schedule; the reference's default sigmoid and the cosine schedule stay at low noise far longer
    # and the same bands cost them 1.0-2.4e-3 (profiles/r04_s_*): they keep three products at every step, switch or no switch
    for sched in ("sigmoid", "cosine"):
        ds = diffusion_class("uncond")(product_unet("uncond", 16, (1, 2)), image_size=16, timesteps=1000, loss_type="l2", beta_schedule=sched).cuda()
        assert ds.early_band_f16 and {ds._coef(t).arith for t in (999, 900, 600, 100, 0)} == {L.HD_ARITH_DEFAULT}
        on, _ = ds.p_sample(x, 900)
        ds.early_band_f16 = False
        off, _ = ds.p_sample(x, 900)
        assert torch.equal(on, off)


@pytest.mark.parametrize("kind", ["uncond", "cond", "sr3"])
def test_two_product_forward_vs_oracle(kind):
    """HD_PRECISION_F16W2 as a context-wide mode (tests and measurements): a full-size UNet forward at 64x64 within the arithmetic's own
    per-forward error of the fp32 oracle (CPU study: 1.1e-3 with every wide convolution rerouted; here the 3x3 ones only)."""
    from hicdiff_amd import _lib as L
    m, ref = product_unet(kind), oracle_unet(kind)
    eng = _eng(m)
    x, t = tiles(41, 2, 64), torch.tensor([700, 30])
    cond = tiles(42, 2, 64) if kind != "uncond" else None
    if kind == "sr3":
        t = torch.tensor([[0.3], [0.9]])
    want = oracle_once(("f16w2", kind), lambda: ref(x, t, cond))
    base = m(x.cuda(), t.cuda(), None if cond is None else cond.cuda())
    eng.set_precision(L.HD_PRECISION_F16W2)
    got = m(x.cuda(), t.cuda(), None if cond is None else cond.cuda())
    eng.set_precision(L.HD_PRECISION_F16W1)
    got1 = m(x.cuda(), t.cuda(), None if cond is None else cond.cuda())
    eng.set_precision(L.HD_PRECISION_BF16X3)
    assert rel_err(want, base) < 1e-4
    assert 1e-5 < rel_err(want, got) < 3e-3
    assert 1e-5 < rel_err(want, got1) < 6e-3 and not torch.equal(got, got1)          # one fp16 product: the weights are rounded as well


def test_fp16_weight_image_range_guard():
    """The loader measures max |w| of every 3x3 layer's packed weights; a layer outside [2^-8, 2^15] -- where fp16 hi | lo (exact to ~2^-25
    absolute) would no longer carry the weight to the accuracy the error budget assumes -- keeps three bf16 products under every arithmetic
    (engine.hip Loader::finish_f16_range).  hicedrn's convolutions are plain nn.Conv2d: scaled by 2^-12 (exact) they fall under the guard and
    the one-product mode must give the x3 result bit for bit; unscaled, the cheaper arithmetic is visible.  Reloading flips the guard back."""
    from _util import product_hicedrn
    from hicdiff_amd import _lib as L
    m = product_hicedrn("uncond", 3)
    x, t = tiles(5, 4, 40).cuda(), torch.tensor([900, 500, 100, 3], device="cuda")

    def both():
        out = []
        for prec in (L.HD_PRECISION_BF16X3, L.HD_PRECISION_F16W1):
            eng = _eng(m)                      # (syncs the weights: the parameters' version counters moved)
            eng.set_precision(prec)
            try:
                out.append(m(x, t).clone())
            finally:
                eng.set_precision(L.HD_PRECISION_BF16X3)
        return out

    a3, a1 = both()
    assert not torch.equal(a3, a1) and rel_err(a3.cpu(), a1) < 6e-3
    wide = [p for p in m.parameters() if p.ndim == 4 and p.shape[-1] == 3 and p.shape[1] % 16 == 0]
    assert len(wide) >= 4
    with torch.no_grad():
        for p in wide:
            p.mul_(2.0 ** -12)
    b3, b1 = both()
    assert torch.equal(b3, b1)
    with torch.no_grad():
        for p in wide:
            p.mul_(2.0 ** 12)
    c3, c1 = both()
    assert torch.equal(c3, a3) and torch.equal(c1, a1)


def test_precision_schedule_added_error_at_bench_batch(capsys):
    """BASELINE's headline workload as `bench.py` runs it -- 256 tiles of 64x64, T = 1000, two half-batch chains, the default precision schedule
    (one / two fp16 products in the early band, two on the low-resolution layers below it) -- against the same chain with split-bf16 x3 at
    every step: what the schedule ADDS, max over all 256 tiles.  Measured 3.5-5.0e-4 over six (flavour, seed) cases (profiles/r04_m_*); the x3
    chain's own distance to the CPU oracle is 1.0-1.6e-4 (test_full_length_chain_drift_vs_oracle and tools/early_band_drift.py), so the bound here
    keeps the sum inside the 1e-3 parity bound."""
    B, S, T = 256, 64, 1000
    net = product_unet("uncond")
    d = diffusion_class("uncond")(net, image_size=S, timesteps=T, loss_type="l2", beta_schedule="linear").cuda()
    d.seed = 1
    assert d.early_band_f16 and d.late_band_low_f16 and _eng(net).chains_for(B, S) == 2
    fast = d.sample(torch.zeros(B, 1, S, S))
    d.early_band_f16 = False
    exact3 = d.sample(torch.zeros(B, 1, S, S))
    added = rel_err(exact3, fast)
    with capsys.disabled():
        print(f"\n[precision schedule, 256 x 64x64, T = 1000] added error {added:.2e}")
    assert 1e-5 < added < 7e-4


def test_full_length_hicedrn_chain_drift_vs_oracle(capsys):
    """hicedrn (8 of its 32 blocks keep the CPU oracle to two minutes), 2 tiles of 40x40, T = 1000: the default path -- device noise, graph
    replay, the precision schedule (hicedrn: one fp16 product for t >= 3T/4, two below) on the 256 -> 256 body convolutions -- against the
    oracle over the same noise."""
    B, S, T, seed = 2, 40, 1000, 404
    net = product_hicedrn("uncond", 8)
    d = diffusion_class("uncond")(net, image_size=S, timesteps=T, loss_type="l2", beta_schedule="linear").cuda()
    d.seed = seed
    assert d._early_band(T - 1) and d._early_band(0)          # hicedrn: the whole chain
    got = d.sample(torch.zeros(B, 1, S, S))
    from oracle import diffusion as OD
    ref = OD.DiffusionRef(oracle_hicedrn("uncond", 8), image_size=S, timesteps=T, beta_schedule="linear", loss_type="l2")
    want = ref.p_sample_loop((B, 1, S, S), AncestralDeviceNoise(B, S, T, seed))
    err = rel_err(want, got)
    d.early_band_f16 = False
    err_x3 = rel_err(want, d.sample(torch.zeros(B, 1, S, S)))
    with capsys.disabled():
        print(f"\n[drift hicedrn8] early band {err:.2e}, split-bf16 x3 at every step {err_x3:.2e}")
    assert err < CHAIN_TOL and err_x3 < CHAIN_TOL


# ---------------------------------------------------------------- bench.py --gpus N as the driver runs it

def test_bench_two_ranks_self_launched_on_one_gpu():
    """`python bench.py --gpus 2` with no launcher: bench.py starts both ranks itself (gloo rendezvous and both ranks on
    GPU 0 here, because the test box has one card; RCCL over two cards is the same code with the default backend)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HICDIFF_BENCH_BACKEND="gloo", HICDIFF_DEVICE="0")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--workload", "unet40",
                          "--batch", "4", "--no-cpu-baseline", "--sustained-budget", "0"], env=env, capture_output=True, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    line = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["scaling"] == "weak"
    assert line["config"]["tiles_per_gpu"] == 4 and line["value"] > 0 and "all_gather_ms" in line
    assert abs(line["value"] - 2 * 4 / (1000 * line["ms_per_step"] * 1e-3)) < 1e-3 * line["value"]


def test_bench_strong_scaling_form_two_ranks():
    """`--total-tiles T` (BASELINE configs[3]'s "256 tiles sharded across the node"): T/N tiles per rank, value = T tiles over the chain time,
    reported as strong scaling; an uneven deal is refused before any GPU work."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HICDIFF_BENCH_BACKEND="gloo", HICDIFF_DEVICE="0")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    base = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "unet40", "--no-cpu-baseline",
            "--sustained-budget", "0"]
    out = subprocess.run(base + ["--total-tiles", "6"], env=env, capture_output=True, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    line = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert line["scaling"] == "strong" and line["n_gpus"] == 2
    assert line["config"]["tiles_per_gpu"] == 3 and line["config"]["total_tiles"] == 6
    assert abs(line["value"] - 6 / (1000 * line["ms_per_step"] * 1e-3)) < 1e-3 * line["value"]
    bad = subprocess.run(base + ["--total-tiles", "7"], env=env, capture_output=True, timeout=120)
    assert bad.returncode != 0 and b"total % gpus" in bad.stderr


def test_bench_training_two_ranks_on_one_gpu():
    """`bench.py --workload hicedrn64_train --gpus 2` (BASELINE configs[4]'s data-parallel form) rehearsed on one card over gloo: both ranks
    step together -- the gradients are summed stage by stage behind the backward pass -- and the line reports the whole job's tiles/s."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HICDIFF_BENCH_BACKEND="gloo", HICDIFF_DEVICE="0")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "hicedrn64_train", "--blocks", "2", "--tile", "16",
                          "--batch", "4", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    line = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak" and line["unit"] == "tiles/s"
    assert abs(line["value"] - 2 * 4 / (line["ms_per_step"] * 1e-3)) < 1e-2 * line["value"]
    assert "stage by stage" in line["config"]["parallelism"]
    assert line["loss_first_last"][1] < line["loss_first_last"][0]
