"""The oracle (oracle/) against every golden vector generated from the reference (CPU, no GPU).

This is the oracle's pin: the fixtures were produced by importing the reference implementation
(tests/golden/make_golden.py); the oracle must reproduce them before any HIP result is trusted.
Schedules are compared bit for bit; network outputs to 2e-5 relative (they agreed exactly in the
generating container; the slack only covers a different BLAS thread count on another host)."""
import json
import os

import numpy as np
import pytest
import torch

from _util import GOLDEN, golden, oracle_hicedrn, oracle_unet, rel_err
from oracle import ddrm as ODR
from oracle import diffusion as OD
from oracle import weights as W

TOL = 2e-5
CHAIN_TOL = 5e-4

FIXTURE_THREADS = 8      # torch threads of the process that wrote tests/golden/*.npz (make_golden.py in the 8-CPU build container)


@pytest.fixture(autouse=True)
def _fixture_thread_count():
    """Every test of this file compares the oracle with reference outputs at 1e-6 .. 5e-4; several of them (DDIM, interpolate) reproduce their
    fixture only when torch's CPU reductions run in the fixture's order, i.e. on its thread count -- whatever the number of cores."""
    n = torch.get_num_threads()
    torch.set_num_threads(FIXTURE_THREADS)
    yield
    torch.set_num_threads(n)


def test_parameter_inventories_match_reference_state_dicts():
    inv = json.load(open(os.path.join(GOLDEN, "param_inventory.json")))
    for kind in ("uncond", "cond", "sr3"):
        mine = W.unet_shapes(self_condition=(kind != "uncond"), sr3=(kind == "sr3"))
        assert [(k, list(s)) for k, s in mine.items()] == [tuple(x) for x in map(tuple, inv["unet_" + kind])]
        mine = W.hicedrn_shapes(self_condition=(kind != "uncond"), sr3=(kind == "sr3"))
        assert [(k, list(s)) for k, s in mine.items()] == [tuple(x) for x in map(tuple, inv["hicedrn_" + kind])]
    assert [k for k, _ in inv["diffusion_buffers"]] == list(OD.BUFFER_NAMES)


def test_closed_form_fill_is_a_pure_function():
    a = W.fill_tensor("downs.0.0.block1.proj.weight", (64, 64, 3, 3))
    b = W.fill_tensor("downs.0.0.block1.proj.weight", (64, 64, 3, 3))
    assert torch.equal(a, b) and abs(a.std().item() - (1 / 576) ** 0.5) < 2e-3
    assert abs(W.fill_tensor("x.norm.weight", (64,)).mean().item() - 1.0) < 0.1
    assert not torch.equal(a, W.fill_tensor("downs.0.1.block1.proj.weight", (64, 64, 3, 3)))


@pytest.mark.parametrize("sched,T", [("linear", 50), ("linear", 1000), ("linear", 2000), ("sigmoid", 1000), ("sigmoid", 2000),
                                     ("cosine", 1000)])
def test_schedule_buffers_bit_exact(sched, T):
    g = golden("schedules")
    mine = OD.diffusion_buffers(sched, T)
    for name in OD.BUFFER_NAMES:
        assert torch.equal(g[f"{sched}_{T}_{name}"], mine[name]), name


def test_sr3_table_and_ddrm_betas_bit_exact():
    g = golden("schedules")
    assert torch.equal(g["linear_2000_sqrt_alphas_cumprod_prev"], OD.diffusion_buffers("linear", 2000)["sqrt_alphas_cumprod_prev"])
    assert torch.equal(g["ddrm_linear_betas"], ODR.ddrm_betas("linear"))


@pytest.mark.parametrize("kind", ["uncond", "cond", "sr3"])
@pytest.mark.parametrize("tag", ["s16", "s40"])
def test_tiny_unet_eps(kind, tag):
    g = golden("tiny")
    pre = f"{kind}_{tag}_"
    out = oracle_unet(kind, 16, (1, 2))(g[pre + "x"], g[pre + "t"], g.get(pre + "cond"))
    assert rel_err(g[pre + "eps"], out) < TOL


def test_tiny_unet_probes():
    from oracle import nets as ON
    g = golden("tiny")
    cfg = ON.UnetCfg(dim=16, dim_mults=(1, 2))
    sd = W.fill_state_dict(W.unet_shapes(dim=16, dim_mults=(1, 2)))
    probes = {}
    with torch.no_grad():
        ON.unet_eps(sd, g["uncond_s16_x"], g["uncond_s16_t"], None, cfg, probes)
    for k in ("init_conv", "time_mlp", "downs.0", "mid", "ups.0"):
        assert rel_err(g["uncond_s16_probe_" + k], probes[k]) < TOL, k


@pytest.mark.parametrize("kind", ["uncond", "cond", "sr3"])
def test_full_unet_eps(kind):
    g = golden("eps")
    m = oracle_unet(kind)
    for tag in ("s40", "s64"):
        pre = f"unet_{kind}_{tag}_"
        assert rel_err(g[pre + "eps"], m(g[pre + "x"], g[pre + "t"], g.get(pre + "cond"))) < TOL
    if kind == "uncond":
        assert rel_err(g["unet_uncond_floatt_eps"], m(g["unet_uncond_floatt_x"], g["unet_uncond_floatt_t"])) < TOL


@pytest.mark.parametrize("kind,nres,s", [("uncond", 3, 64), ("cond", 3, 64), ("sr3", 3, 40), ("uncond", 32, 40)])
def test_hicedrn_eps(kind, nres, s):
    g = golden("eps")
    pre = f"hicedrn_{kind}_n{nres}_s{s}_"
    out = oracle_hicedrn(kind, nres)(g[pre + "x"], g[pre + "t"], g.get(pre + "cond"))
    assert rel_err(g[pre + "eps"], out) < TOL


def test_ancestral_chains():
    g = golden("trajectories")
    ref = OD.DiffusionRef(oracle_unet("uncond"), image_size=40, timesteps=50, beta_schedule="linear", loss_type="l2")
    _, kept = ref.p_sample_loop((2, 1, 40, 40), OD.TorchNoise(1234), keep_every=10)
    assert torch.equal(kept[50], g["uncond_xT"])
    for t in range(0, 50, 10):
        assert rel_err(g[f"uncond_x_after_t{t}"], kept[t]) < CHAIN_TOL
    for kind, seed in (("cond", 4321), ("sr3", 999)):
        ref = OD.DiffusionRef(oracle_unet(kind), image_size=40, timesteps=50, beta_schedule="linear", loss_type="l2", kind=kind)
        _, kept = ref.p_sample_loop(g["cond_lq"], OD.TorchNoise(seed), keep_every=10)
        for t in range(0, 50, 10):
            assert rel_err(g[f"{kind}_x_after_t{t}"], kept[t]) < CHAIN_TOL


def test_ddim():
    """The fixture was written by the reference on 8 torch threads and the oracle reproduces it bit for bit on 8 -- on 6 it ends 8.0e-4 away, on 4
    1.0e-3 (and so would the reference from its own fixture): torch's CPU reductions follow the thread count and a 20-step DDIM chain on the
    sigmoid schedule amplifies the reordering a thousandfold.  The thread COUNT, not the core count, fixes the order: `_fixture_thread_count` pins it."""
    g = golden("trajectories")
    for eta in (0.0, 0.5):
        ref = OD.DiffusionRef(oracle_unet("uncond"), image_size=40, timesteps=1000, beta_schedule="sigmoid",
                              sampling_timesteps=20, ddim_sampling_eta=eta)
        assert rel_err(g[f"ddim_eta{eta}_x0"], ref.ddim_sample((2, 1, 40, 40), OD.TorchNoise(55))) < CHAIN_TOL


@pytest.mark.parametrize("net", ["unet", "hicedrn3"])
def test_ddrm_chains(net):
    g = golden("trajectories")
    model = oracle_unet("uncond") if net == "unet" else oracle_hicedrn("uncond", 3)
    betas = ODR.ddrm_betas("linear")
    for sigma_0 in (0.1, 1.0):
        pre = f"ddrm_{net}_s{sigma_0}_"
        nz = OD.TorchNoise(2024)
        x = nz.randn((2, 1, 40, 40))
        out, x0, kept = ODR.ddrm_denoise(x, range(0, 1000, 20), model, betas, g[pre + "y0"], sigma_0, noise=nz, keep_steps=(10, 25, 40))
        for k in (10, 25, 40):
            assert rel_err(g[pre + f"x_step{k}"], kept[k]) < CHAIN_TOL
        assert rel_err(g[pre + "final"], out) < CHAIN_TOL
        assert rel_err(g[pre + "x0_last"], x0) < CHAIN_TOL


def test_losses_and_q_sample():
    g = golden("losses")
    for loss in ("l1", "l2"):
        ref = OD.DiffusionRef(oracle_unet("uncond"), image_size=40, timesteps=1000, beta_schedule="sigmoid", loss_type=loss)
        val = ref.p_losses(g["x0"], g[f"uncond_{loss}_t"], g[f"uncond_{loss}_eps"])
        assert abs(val.item() - g[f"uncond_{loss}_loss"].item()) < 1e-5 * abs(val.item())
    ref = OD.DiffusionRef(oracle_unet("cond"), image_size=40, timesteps=1000, beta_schedule="linear", loss_type="l2", kind="cond")
    val = ref.p_losses(g["x0"], g["cond_l2_t"], g["cond_l2_eps"], g["lq"])
    assert abs(val.item() - g["cond_l2_loss"].item()) < 1e-5 * abs(val.item())
    ref = OD.DiffusionRef(oracle_unet("sr3"), image_size=40, timesteps=2000, beta_schedule="linear", loss_type="l2", kind="sr3")
    assert torch.equal(ref.sr3_draw_level(np.random.RandomState(7), 4), g["sr3_l2_level"])
    val = ref.p_losses_sr3(g["x0"], g["sr3_l2_level"], g["sr3_l2_eps"], g["lq"])
    assert abs(val.item() - g["sr3_l2_loss"].item()) < 1e-5 * abs(val.item())
    ref = OD.DiffusionRef(None, image_size=40, timesteps=1000, beta_schedule="sigmoid")
    assert torch.equal(ref.q_sample(g["x0"], g["q_sample_t"], g["uncond_l2_eps"]), g["q_sample_out"])


def test_metrics_oracle_reproduces_reference_ssim_and_batch_values():
    """oracle/metrics.py against the values the reference's own SSIM module and formulas produced (make_golden.py)."""
    from oracle import metrics as OM
    g = np.load(os.path.join(GOLDEN, "metrics.npz"))
    assert torch.equal(OM.create_window(11, 1), torch.from_numpy(g["window"]))
    for tag in ("s40", "s64", "far"):
        pr, hq = torch.from_numpy(g[f"{tag}_pred"]), torch.from_numpy(g[f"{tag}_target"])
        o, h = OM.rescaled(pr), OM.rescaled(hq)
        assert abs(float(OM.ssim(o, h)) - float(g[f"{tag}_ssim"])) < 1e-7
        assert np.abs(OM.ssim(o, h, size_average=False).numpy() - g[f"{tag}_ssim_each"]).max() < 1e-7
        m = OM.batch_metrics(pr, hq)
        assert abs(m["mse"] - float(g[f"{tag}_mse"])) <= 1e-6 * float(g[f"{tag}_mse"])
        assert abs(m["snr"] - float(g[f"{tag}_snr"])) <= 1e-5 * abs(float(g[f"{tag}_snr"]))
        assert abs(m["pcc"] - float(g[f"{tag}_pcc"])) < 1e-6
        assert abs(m["psnr"] - float(g[f"{tag}_psnr"])) < 1e-4


TILE_CASES = ["pad40", "exact64", "band8", "res10k", "small", "gap", "overlap"]


@pytest.mark.parametrize("tag", TILE_CASES)
def test_tiles_oracle_reproduces_reference_split(tag):
    """oracle/tiles.py against the arrays the reference's own splitPieces cut (make_golden.py::case_tiles), bit for bit;
    the host side's tile_origins (pure integer logic, no GPU) must list the same tiles in the same order."""
    from oracle import tiles as OT
    from hicdiff_amd.processdata import tile_origins
    g = np.load(os.path.join(GOLDEN, "tiles.npz"))
    n, p, st, res = (int(v) for v in g[f"{tag}_args"])
    mine = OT.split_pieces(g[f"{tag}_mat"], p, st, res)
    assert mine.dtype == g[f"{tag}_tiles"].dtype and np.array_equal(mine, g[f"{tag}_tiles"])
    org, bound = tile_origins(n, p, st, res)
    assert np.array_equal(org, g[f"{tag}_origins"]) and bound % p == 0 and bound - n < p
    if st >= p:                                                    # the stitch definition: a symmetric map comes back inside the band
        back = OT.stitch_pieces(mine, org, n)
        held = OT.stitch_pieces(np.ones_like(mine), org, n) > 0
        assert np.array_equal(back[held], g[f"{tag}_mat"][held]) and not back[~held].any()
        assert np.array_equal(back, back.T)


def test_tiles_edge_cases_and_degradation():
    from oracle import tiles as OT
    from hicdiff_amd.processdata import tile_origins
    g = np.load(os.path.join(GOLDEN, "tiles.npz"))
    assert tuple(g["empty_shape"]) == (0, 1) == OT.split_pieces(np.zeros((0, 0), np.float32), 8, 8, 40000).shape
    assert len(tile_origins(0, 8, 8, 40000)[0]) == 0
    n, p, st, res = (int(v) for v in g["ragged_raises"])
    with pytest.raises(ValueError):
        OT.split_pieces(np.zeros((n, n), np.float32), p, st, res)
    with pytest.raises(ValueError):
        tile_origins(n, p, st, res)
    noisy, sample = OT.degrade(g["pad40_tiles"], 0.1, g["deg_z"])
    assert np.array_equal(noisy, g["deg_noisy"]) and np.array_equal(sample, g["deg_sample"])


@pytest.mark.parametrize("kind", ["cond", "uncond", "sr3"])
def test_train_oracle_reproduces_reference_gradients_and_adam(kind):
    """oracle/train.py (autograd over the oracle net + the Adam restatement) against what the reference's own
    `loss.backward(); torch.optim.Adam.step()` produced for three steps (make_golden.py::case_train)."""
    from oracle import diffusion as OD, nets as ON, train as OTR, weights as W
    g = np.load(os.path.join(GOLDEN, "train.npz"))
    cfg = ON.HicedrnCfg(number_resnet=2, self_condition=(kind != "uncond"), sr3=(kind == "sr3"))
    sd = W.fill_state_dict(W.hicedrn_shapes(number_resnet=2, self_condition=cfg.self_condition, sr3=cfg.sr3))
    buf = OD.diffusion_buffers("linear", 1000)
    m, v = {k: torch.zeros_like(p) for k, p in sd.items()}, {k: torch.zeros_like(p) for k, p in sd.items()}
    x0, lq = torch.from_numpy(g["x0"]), torch.from_numpy(g["lq"])
    for step in (1, 2, 3):
        t, eps = torch.from_numpy(g[f"{kind}_s{step}_t"]), torch.from_numpy(g[f"{kind}_s{step}_eps"])
        loss, grads = OTR.loss_and_grads(sd, cfg, buf, x0, t, eps, None if kind == "uncond" else lq, "l2")
        assert abs(float(loss) - float(g[f"{kind}_s{step}_loss"])) <= 1e-6 * float(loss)
        for k in sd:
            ref = torch.from_numpy(g[f"{kind}_s{step}_grad_sample/{k}"])
            assert float((OTR.sample_of(grads[k]) - ref).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1e-12), (step, k)
            assert abs(float(grads[k].norm()) - float(g[f"{kind}_s{step}_grad_norm/{k}"])) <= 1e-5 * float(grads[k].norm())
        OTR.adam_step(sd, grads, m, v, step)
        for k in sd:
            ref = torch.from_numpy(g[f"{kind}_s{step}_param_sample/{k}"])
            assert float((OTR.sample_of(sd[k]) - ref).abs().max()) <= 1e-6 * float(ref.abs().max()) + 1e-9, (step, k)


@pytest.mark.parametrize("kind", ["cond", "uncond", "sr3"])
def test_train_oracle_reproduces_reference_unet_gradients(kind):
    """oracle/train.py on the two-level UNet against the reference's own loss.backward() (make_golden.py::case_train_unet), first step."""
    from oracle import diffusion as OD, nets as ON, train as OTR, weights as W
    g = np.load(os.path.join(GOLDEN, "train_unet.npz"))
    cfg = ON.UnetCfg(dim=64, dim_mults=(1, 2), self_condition=(kind != "uncond"), sr3=(kind == "sr3"))
    sd = W.fill_state_dict(W.unet_shapes(dim=64, dim_mults=(1, 2), self_condition=cfg.self_condition, sr3=cfg.sr3))
    x0, lq = torch.from_numpy(g["x0"]), torch.from_numpy(g["lq"])
    t, eps = torch.from_numpy(g[f"{kind}_s1_t"]), torch.from_numpy(g[f"{kind}_s1_eps"])
    loss, grads = OTR.loss_and_grads(sd, cfg, OD.diffusion_buffers("linear", 1000), x0, t, eps, None if kind == "uncond" else lq, "l2")
    assert abs(float(loss) - float(g[f"{kind}_s1_loss"])) <= 1e-6 * float(loss)
    for k in sd:
        ref = torch.from_numpy(g[f"{kind}_s1_grad_sample/{k}"])
        assert float((OTR.sample_of(grads[k]) - ref).abs().max()) <= 5e-5 * max(float(ref.abs().max()), 1e-12), k


GENERAL_CHAINS = ("inp_mask", "sr2", "deblur_uni", "deblur_gauss", "deblur_aniso", "cs2")


@pytest.mark.parametrize("deg", GENERAL_CHAINS)
def test_ddrm_general_chain_golden(deg):
    """the oracle's general DDRM sampler (operator given as the dense matrices the reference's own V / U
    methods produce) reproduces the reference's chains on replayed noise."""
    from oracle import ddrm as ODR, diffusion as OD
    g = golden("ddrm_general")
    pre = f"{deg}_c1_"
    H = ODR.DenseH(g[pre + "V"].T, g[pre + "U"].T, g[pre + "s"])
    model = oracle_unet("uncond", 16, (1, 2))
    nz = OD.TorchNoise(2025)
    x = nz.randn((2, 1, 8, 8))
    out, x0 = ODR.ddrm_general(x, range(0, 1000, 100), model, ODR.ddrm_betas("linear"), H, g[pre + "y0"], 0.1, noise=nz)
    assert rel_err(g[pre + "final"], out) < CHAIN_TOL
    assert rel_err(g[pre + "x0_last"], x0) < CHAIN_TOL
    # V is orthogonal and V^T its transpose, as the sampler assumes
    V = g[pre + "V"]
    assert torch.allclose(V @ V.T, torch.eye(V.shape[0]), atol=2e-5) and torch.allclose(g[pre + "Vt"], V.T, atol=2e-5)


def test_oracle_objectives_pred_x0_and_pred_v_reproduce_the_reference():
    """objective = 'pred_x0' / 'pred_v' (src/hicdiff.py:566-580,733-741): 20-step ancestral chain, 5-of-50 DDIM (eta 0.5) and the loss value, against
    fixtures the reference produced (tests/golden/make_golden.py::case_objectives)."""
    from _util import oracle_unet
    from oracle import diffusion as OD
    g = golden("objectives")
    model = oracle_unet("uncond")
    for obj in ("pred_x0", "pred_v"):
        ref = OD.DiffusionRef(model, image_size=40, timesteps=20, beta_schedule="linear", loss_type="l2", objective=obj)
        final, kept = ref.p_sample_loop((2, 1, 40, 40), OD.TorchNoise(4242), keep_every=5)
        for k in range(0, 20, 5):
            assert rel_err(g[f"{obj}_x_after_t{k}"], kept[k]) < 1e-5, (obj, k)
        assert abs(float(ref.p_losses(g["x0"], g["t"], g["eps"])) - float(g[f"{obj}_loss"])) <= 1e-5 * float(g[f"{obj}_loss"])
        ddim = OD.DiffusionRef(model, image_size=40, timesteps=50, beta_schedule="linear", sampling_timesteps=5, ddim_sampling_eta=0.5, objective=obj)
        assert rel_err(g[f"{obj}_ddim_x0"], ddim.ddim_sample((2, 1, 40, 40), OD.TorchNoise(77))) < 1e-5


def test_oracle_interpolate_reproduces_the_reference():
    """GaussianDiffusion.interpolate (src/hicdiff.py:673-691): default (t = T - 1, lam = 0.5) and t = 20, lam = 0.3."""
    from oracle import diffusion as OD
    g = golden("interpolate")
    ref = OD.DiffusionRef(oracle_unet("uncond"), image_size=40, timesteps=50, beta_schedule="linear", loss_type="l2")
    for tag, t, lam, seed in (("default", None, 0.5, 611), ("t20_lam03", 20, 0.3, 612)):
        got = ref.interpolate(g["x1"], g["x2"], OD.TorchNoise(seed), t=t, lam=lam)
        assert rel_err(g[tag], got) < 1e-6, tag
