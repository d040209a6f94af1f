"""Native training step against the reference-derived fixtures and the autograd oracle.

Tolerances (written here, fp32 gradients through split-bf16 x3 MFMA products): every gradient tensor within 1e-3 of its own
largest reference entry (max|d| / max|ref|); losses within 1e-4 relative; parameters after Adam steps within 2e-3 of lr per
entry (Adam's update is ~lr * sign-like, so a gradient error can move an update by a fraction of lr, never more than 2 lr)."""
import numpy as np
import pytest
import torch

from _util import golden, product_hicedrn, rel_err, tiles

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("net,loss", [("hicedrn", "l2"), ("unet", "l1")])
def test_train_p2_loss_weight_vs_autograd_oracle(net, loss):
    """GaussianDiffusion(p2_loss_weight_gamma != 0) (src/hicdiff.py:443,522,746; unused by the reference's drivers, but a constructor argument
    of the kept API): the per-sample weights reach the native step's loss gradient (hd_train_set_loss_weights).  Loss and every gradient
    against torch autograd over the oracle net with the same weights; gamma = 0 afterwards restores the plain mean."""
    from oracle import diffusion as OD, nets as ON, train as OTR, weights as W
    from hicdiff_amd.hicdiff import GaussianDiffusion
    B, S, gamma, k = 4, 16, 1.0, 1.0
    if net == "hicedrn":
        m = product_hicedrn("uncond", 2)
        sd, cfg = _oracle_sd("uncond", 2)
    else:
        from _util import product_unet
        m = product_unet("uncond", dim=64, mults=(1, 2))
        cfg = ON.UnetCfg(dim=64, dim_mults=(1, 2), self_condition=False, sr3=False)
        sd = W.fill_state_dict(W.unet_shapes(dim=64, dim_mults=(1, 2), self_condition=False, sr3=False))
    d = GaussianDiffusion(m, image_size=S, timesteps=1000, loss_type=loss, beta_schedule="sigmoid", p2_loss_weight_gamma=gamma, p2_loss_weight_k=k).cuda()
    d.train()
    x0 = tiles(51, B, S)
    gen = torch.Generator().manual_seed(9)
    t, eps = torch.tensor([3, 250, 600, 990]), torch.randn(x0.shape, generator=gen)
    buf = OD.diffusion_buffers("sigmoid", 1000, p2_gamma=gamma, p2_k=k)
    assert torch.equal(buf["p2_loss_weight"], d.p2_loss_weight.cpu()) and float(buf["p2_loss_weight"].min()) < 0.1
    ol, og = OTR.loss_and_grads(sd, cfg, buf, x0, t, eps, None, loss)
    ol1, _ = OTR.loss_and_grads(sd, cfg, OD.diffusion_buffers("sigmoid", 1000), x0, t, eps, None, loss)
    assert abs(float(ol) - float(ol1)) > 0.05 * float(ol1)              # (the weights really change the loss)
    val = d.p_losses(x0.cuda(), t.cuda(), eps.cuda())
    val.backward()
    assert abs(float(val.detach()) - float(ol)) <= 1e-4 * float(ol)
    bad = {kk: rel_err(og[kk], p.grad) for kk, p in d.model.named_parameters() if not rel_err(og[kk], p.grad) <= 1e-3}
    assert not bad, bad
    # the same network under a diffusion object without weights: the trainer goes back to the plain mean
    d1 = GaussianDiffusion(m, image_size=S, timesteps=1000, loss_type=loss, beta_schedule="sigmoid").cuda()
    d1.train()
    for p in m.parameters():
        p.grad = None
    v1 = d1.p_losses(x0.cuda(), t.cuda(), eps.cuda())
    assert abs(float(v1.detach()) - float(ol1)) <= 1e-4 * float(ol1)


def _diffusion(kind, nres, S, loss="l2", schedule="linear"):
    m = product_hicedrn(kind, nres)
    if kind == "cond":
        from hicdiff_amd.hicdiff_condition import GaussianDiffusion
    elif kind == "sr3":
        from hicdiff_amd.hicdiff_sr3 import GaussianDiffusion
    else:
        from hicdiff_amd.hicdiff import GaussianDiffusion
    return GaussianDiffusion(m, image_size=S, timesteps=2000 if kind == "sr3" else 1000, loss_type=loss, beta_schedule=schedule).cuda()


def _oracle_sd(kind, nres):
    from oracle import nets as ON, weights as W
    cfg = ON.HicedrnCfg(number_resnet=nres, self_condition=(kind != "uncond"), sr3=(kind == "sr3"))
    return W.fill_state_dict(W.hicedrn_shapes(number_resnet=nres, self_condition=cfg.self_condition, sr3=cfg.sr3)), cfg


def _loss(d, kind, lq, x0, t, eps):
    """p_losses with given draws: integer timesteps, or (SR3) the continuous noise level."""
    if kind == "sr3":
        return d.p_losses([lq, x0], noise=eps, level=t)
    return d.p_losses([lq, x0], t, eps) if kind == "cond" else d.p_losses(x0, t, eps)


@pytest.mark.parametrize("kind", ["cond", "uncond", "sr3"])
def test_train_three_steps_golden(kind):
    """loss = diffusion(x); loss.backward(); Adam(lr=2e-5).step() -- three steps on the 2-block hicedrn the reference ran
    (tests/golden/make_golden.py::case_train): losses, a fixed subset + the norm of every gradient, parameters after each step."""
    from hicdiff_amd.optim import Adam
    from oracle.train import sample_of
    g = golden("train")
    d = _diffusion(kind, 2, 16)
    d.train()
    opt = Adam(d.parameters(), lr=2e-5)
    x0, lq = g["x0"].cuda(), g["lq"].cuda()
    names = [k for k, _ in d.model.named_parameters()]
    start = {k: p.detach().clone() for k, p in d.model.named_parameters()}
    for step in (1, 2, 3):
        t, eps = g[f"{kind}_s{step}_t"].cuda(), g[f"{kind}_s{step}_eps"].cuda()
        loss = _loss(d, kind, lq, x0, t, eps)
        assert loss.requires_grad
        loss.backward()
        assert abs(float(loss.detach()) - float(g[f"{kind}_s{step}_loss"])) <= 1e-4 * float(g[f"{kind}_s{step}_loss"])
        for k, p in d.model.named_parameters():
            ref_s, ref_n = g[f"{kind}_s{step}_grad_sample/{k}"], float(g[f"{kind}_s{step}_grad_norm/{k}"])
            got = p.grad.detach().cpu()
            assert abs(float(got.norm()) - ref_n) <= 1e-3 * ref_n + 1e-12, (step, k)
            scale = max(float(ref_s.abs().max()), ref_n / got.numel() ** 0.5)
            assert float((sample_of(got) - ref_s).abs().max()) <= 1e-3 * scale, (step, k)
        opt.step()
        opt.zero_grad()
        assert all(p.grad is None for p in d.model.parameters())
        for k, p in d.model.named_parameters():
            ref = g[f"{kind}_s{step}_param_sample/{k}"]
            got = sample_of(p.detach().cpu())
            assert float((got - ref).abs().max()) <= 2e-3 * 2e-5 * step + 1e-7 * float(ref.abs().max()), (step, k)
    moved = max(float((p.detach() - start[k]).abs().max()) for k, p in d.model.named_parameters())
    assert 1e-5 < moved < 1e-4                                  # three Adam steps of lr 2e-5
    assert list(d.model.state_dict().keys()) == names           # checkpoint keys untouched by the flat re-seating


@pytest.mark.parametrize("kind,loss,B,S,nres", [("cond", "l2", 4, 40, 3), ("uncond", "l1", 2, 64, 2), ("cond", "l2", 5, 24, 1), ("sr3", "l2", 3, 40, 2),
                                                 ("sr3", "l1", 2, 16, 1)])
def test_train_gradients_vs_autograd_oracle(kind, loss, B, S, nres):
    """Every entry of every gradient against torch autograd over the oracle net (CPU fp32)."""
    from oracle import diffusion as OD, train as OTR
    d = _diffusion(kind, nres, S, loss=loss, schedule="sigmoid")
    d.train()
    sd, cfg = _oracle_sd(kind, nres)
    x0, lq = tiles(31, B, S), tiles(32, B, S)
    gen = torch.Generator().manual_seed(5)
    t = torch.rand((B,), generator=gen) * 0.9 + 0.05 if kind == "sr3" else torch.randint(0, 1000, (B,), generator=gen)
    eps = torch.randn(x0.shape, generator=gen)
    ol, og = OTR.loss_and_grads(sd, cfg, OD.diffusion_buffers("sigmoid", 1000), x0, t, eps, None if kind == "uncond" else lq, loss)
    val = _loss(d, kind, lq.cuda(), x0.cuda(), t.cuda(), eps.cuda())
    val.backward()
    assert abs(float(val.detach()) - float(ol)) <= 1e-4 * float(ol)
    worst = {}
    for k, p in d.model.named_parameters():
        worst[k] = rel_err(og[k], p.grad)
    bad = {k: v for k, v in worst.items() if not v <= 1e-3}
    assert not bad, bad
    # deterministic: the same call repeats bit for bit
    first = {k: p.grad.clone() for k, p in d.model.named_parameters()}
    for p in d.model.parameters():
        p.grad = None
    val2 = _loss(d, kind, lq.cuda(), x0.cuda(), t.cuda(), eps.cuda())
    val2.backward()
    assert torch.equal(val, val2) and all(torch.equal(first[k], p.grad) for k, p in d.model.named_parameters())


@pytest.mark.parametrize("net,objective,loss", [("hicedrn", "pred_x0", "l2"), ("hicedrn", "pred_v", "l1"), ("unet", "pred_v", "l2")])
def test_train_objectives_pred_x0_and_pred_v_vs_autograd_oracle(net, objective, loss):
    """GaussianDiffusion(objective = 'pred_x0' | 'pred_v') trains natively too (hd_train_set_objective): the loss gradient is taken against
    x_start or v = a_t eps - s_t x_start (src/hicdiff.py:733-741) instead of the noise.  Loss and every gradient against torch autograd over
    the oracle net; the oracle's loss for these objectives is pinned by fixtures the reference produced (tests/golden/objectives.npz)."""
    from oracle import diffusion as OD, nets as ON, train as OTR, weights as W
    from hicdiff_amd.hicdiff import GaussianDiffusion
    B, S = 3, 16
    if net == "hicedrn":
        m = product_hicedrn("uncond", 2)
        sd, cfg = _oracle_sd("uncond", 2)
    else:
        from _util import product_unet
        m = product_unet("uncond", dim=64, mults=(1, 2))
        cfg = ON.UnetCfg(dim=64, dim_mults=(1, 2), self_condition=False, sr3=False)
        sd = W.fill_state_dict(W.unet_shapes(dim=64, dim_mults=(1, 2), self_condition=False, sr3=False))
    d = GaussianDiffusion(m, image_size=S, timesteps=1000, loss_type=loss, beta_schedule="sigmoid", objective=objective).cuda()
    d.train()
    x0 = tiles(41, B, S)
    gen = torch.Generator().manual_seed(8)
    t, eps = torch.randint(0, 1000, (B,), generator=gen), torch.randn(x0.shape, generator=gen)
    ol, og = OTR.loss_and_grads(sd, cfg, OD.diffusion_buffers("sigmoid", 1000), x0, t, eps, None, loss, objective)
    ol0, _ = OTR.loss_and_grads(sd, cfg, OD.diffusion_buffers("sigmoid", 1000), x0, t, eps, None, loss)
    assert abs(float(ol) - float(ol0)) > 1e-3 * float(ol0)            # (a different target really gives a different loss)
    val = d.p_losses(x0.cuda(), t.cuda(), eps.cuda())
    val.backward()
    assert abs(float(val.detach()) - float(ol)) <= 1e-4 * float(ol)
    bad = {k: rel_err(og[k], p.grad) for k, p in d.model.named_parameters() if not rel_err(og[k], p.grad) <= 1e-3}
    assert not bad, bad


def test_train_loop_matches_oracle_loss_curve_and_serves_updated_weights():
    """Twenty steps of the reference loop (forward draws t and noise itself): the loss curve follows the oracle's Adam run on the
    same draws; afterwards eval-mode sampling uses the UPDATED weights (the inference engine re-packs), and a state_dict round trip
    through a fresh module reproduces eps."""
    from hicdiff_amd.optim import Adam
    from oracle import diffusion as OD, nets as ON, train as OTR
    B, S, nres = 4, 16, 2
    d = _diffusion("cond", nres, S)
    d.train()
    opt = Adam(d.parameters(), lr=2e-4)
    sd, cfg = _oracle_sd("cond", nres)
    buf = OD.diffusion_buffers("linear", 1000)
    om, ov = {k: torch.zeros_like(v) for k, v in sd.items()}, {k: torch.zeros_like(v) for k, v in sd.items()}
    x0, lq = tiles(41, B, S).cuda(), tiles(42, B, S).cuda()
    before = d.model(x0, torch.full((B,), 10, device="cuda"), lq).clone()
    for step in range(1, 21):
        torch.manual_seed(500 + step)
        loss = d([lq, x0])
        loss.backward()
        opt.step()
        opt.zero_grad()
        torch.manual_seed(500 + step)
        t = torch.randint(0, 1000, (B,), device="cuda").long()
        eps = torch.randn_like(x0)
        ol, og = OTR.loss_and_grads(sd, cfg, buf, x0.cpu(), t.cpu(), eps.cpu(), lq.cpu(), "l2")
        OTR.adam_step(sd, og, om, ov, step, lr=2e-4)
        assert abs(float(loss.detach()) - float(ol)) <= 2e-3 * float(ol), step
    d.eval()
    with torch.no_grad():
        after = d.model(x0, torch.full((B,), 10, device="cuda"), lq)
        ref = ON.hicedrn_eps(sd, x0.cpu(), torch.full((B,), 10), lq.cpu(), cfg)
    assert rel_err(ref, after) < 2e-3 and rel_err(before, after) > 1e-3
    fresh = product_hicedrn("cond", nres)
    fresh.load_state_dict(d.model.state_dict())
    with torch.no_grad():
        assert torch.equal(fresh(x0, torch.full((B,), 10, device="cuda"), lq), after)


def test_train_full_network_properties():
    """Size-independent properties on the full 32-block network at 64x64 tiles: the step repeats bit for bit, and the batch
    gradient is the mean of its halves' gradients (the loss is a mean over samples; tiles never interact)."""
    d = _diffusion("cond", 32, 64)
    d.train()
    B = 8
    x0, lq = tiles(81, B, 64).cuda(), tiles(82, B, 64).cuda()
    gen = torch.Generator().manual_seed(9)
    t, eps = torch.randint(0, 1000, (B,), generator=gen).cuda(), torch.randn(x0.shape, generator=gen).cuda()

    def grads(sl):
        for p in d.model.parameters():
            p.grad = None
        loss = d.p_losses([lq[sl], x0[sl]], t[sl], eps[sl])
        loss.backward()
        return float(loss.detach()), {k: p.grad.clone() for k, p in d.model.named_parameters()}

    l_all, g_all = grads(slice(0, B))
    l_again, g_again = grads(slice(0, B))
    assert l_all == l_again and all(torch.equal(g_all[k], g_again[k]) for k in g_all)
    l_a, g_a = grads(slice(0, B // 2))
    l_b, g_b = grads(slice(B // 2, B))
    assert abs(l_all - 0.5 * (l_a + l_b)) <= 1e-5 * l_all
    for k in g_all:
        assert rel_err(g_all[k], 0.5 * (g_a[k] + g_b[k])) <= 2e-4, k
    assert all(torch.isfinite(v).all() for v in g_all.values())


def test_train_ragged_batch_keeps_adam_state():
    """A smaller last batch re-sizes the trainer (new saved-activation buffers, parameters re-seated); Adam's moments and step
    count belong to the network and carry over: the run follows the oracle's Adam run on the same draws, not one whose state
    was reset when the batch size changed."""
    from hicdiff_amd.optim import Adam
    from oracle import diffusion as OD, train as OTR
    d = _diffusion("cond", 1, 16)
    d.train()
    opt = Adam(d.parameters(), lr=1e-3)
    sd, cfg = _oracle_sd("cond", 1)
    buf = OD.diffusion_buffers("linear", 1000)
    zeros = lambda: {k: torch.zeros_like(v) for k, v in sd.items()}
    keep, (km, kv) = {k: v.clone() for k, v in sd.items()}, (zeros(), zeros())
    reset, (rm, rv), rstep = {k: v.clone() for k, v in sd.items()}, (zeros(), zeros()), 0
    for step, B in enumerate((4, 4, 3, 4), start=1):
        x0, lq = tiles(60 + step, B, 16), tiles(70 + step, B, 16)
        gen = torch.Generator().manual_seed(step)
        t, eps = torch.randint(0, 1000, (B,), generator=gen), torch.randn(x0.shape, generator=gen)
        loss = d.p_losses([lq.cuda(), x0.cuda()], t.cuda(), eps.cuda())
        loss.backward()
        opt.step()
        opt.zero_grad()
        ol, og = OTR.loss_and_grads(keep, cfg, buf, x0, t, eps, lq, "l2")
        OTR.adam_step(keep, og, km, kv, step, lr=1e-3)
        assert abs(float(loss.detach()) - float(ol)) <= 2e-3 * float(ol), step
        if B != 4 or step == 4:                             # what a per-trainer state would do: forget the moments at every re-size
            (rm, rv), rstep = (zeros(), zeros()), 0
        rstep += 1
        _, rg = OTR.loss_and_grads(reset, cfg, buf, x0, t, eps, lq, "l2")
        OTR.adam_step(reset, rg, rm, rv, rstep, lr=1e-3)
    dist = lambda ref: sum(float((p.detach().cpu() - ref[k]).pow(2).sum()) for k, p in d.model.named_parameters()) ** 0.5
    assert dist(keep) < 0.25 * dist(reset), (dist(keep), dist(reset))


def test_train_gradient_accumulation_over_micro_batches():
    """Two diffusion(x) / backward() pairs with no zero_grad between leave the SUM of both micro-batches' gradients in .grad
    (torch's AccumulateGrad semantics), although the kernels overwrite the flat gradient buffer the .grad views alias."""
    d = _diffusion("cond", 1, 16)
    d.train()
    draws = []
    for k in range(2):
        gen = torch.Generator().manual_seed(40 + k)
        draws.append((tiles(80 + k, 3, 16).cuda(), tiles(90 + k, 3, 16).cuda(), torch.randint(0, 1000, (3,), generator=gen).cuda(),
                      torch.randn((3, 1, 16, 16), generator=gen).cuda()))
    singles = []
    for lq, x0, t, eps in draws:
        d.p_losses([lq, x0], t, eps).backward()
        singles.append({n: p.grad.clone() for n, p in d.model.named_parameters()})
        for p in d.parameters():
            p.grad = None
    for lq, x0, t, eps in draws:
        d.p_losses([lq, x0], t, eps).backward()               # no zero_grad in between
    for n, p in d.model.named_parameters():
        want = singles[0][n] + singles[1][n]
        assert torch.allclose(p.grad, want, rtol=0, atol=1e-6 * float(want.abs().max()) + 1e-12), n
    # a third micro-batch keeps accumulating, and zero_grad starts over
    lq, x0, t, eps = draws[0]
    d.p_losses([lq, x0], t, eps).backward()
    n0, p0 = next(iter(d.model.named_parameters()))
    want = 2 * singles[0][n0] + singles[1][n0]
    assert torch.allclose(p0.grad, want, rtol=0, atol=2e-6 * float(want.abs().max()))
    for p in d.parameters():
        p.grad = None
    d.p_losses([lq, x0], t, eps).backward()
    assert torch.equal(p0.grad, singles[0][n0])


def test_train_errors():
    from hicdiff_amd.optim import Adam
    from _util import product_unet
    from hicdiff_amd.hicdiff import GaussianDiffusion
    u = GaussianDiffusion(product_unet("uncond", dim=16, mults=(1, 2)), image_size=16, timesteps=50).cuda()
    u.train()
    loss = u(tiles(1, 2, 16).cuda())                      # UNet: loss value only, as before (no native backward yet)
    assert not loss.requires_grad
    with pytest.raises(NotImplementedError):
        Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-3, weight_decay=0.1)
    d = _diffusion("uncond", 1, 16)
    d.train()
    first = d(tiles(1, 2, 16).cuda())
    second = d(tiles(2, 2, 16).cuda())
    with pytest.raises(RuntimeError):
        first.backward()                                  # its gradients were overwritten by the second forward
    second.backward()
    with pytest.raises(AssertionError):
        d(tiles(1, 2, 24).cuda())                         # wrong tile size: the reference's assert


def test_train_cli_trains_and_writes_reference_checkpoints(tmp_path, capsys):
    """train.py end to end (2-block hicedrn, conditional, synthetic tiles): the loss falls, bestg_/finalg_ files carry the
    reference's names (train.py:185,189) and load into a fresh module with strict keys."""
    import json
    import train
    best = train.main(["-u", "", "-b", "4", "-e", "4", "--resnet-blocks", "2", "--tile", "16", "--tiles-per-epoch", "16", "--lr", "5e-4",
                       "--weights-dir", str(tmp_path)])
    lines = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert [l["Epoch"] for l in lines] == [1, 2, 3, 4]
    assert lines[-1]["train/loss"] < 0.8 * lines[0]["train/loss"] and best == min(l["valid/loss"] for l in lines)
    names = sorted(p.name for p in tmp_path.iterdir())
    assert names == ["bestg_40000_c16_s16_Human1_HiCedrn_cond_l2_lin.pytorch", "finalg_40000_c16_s16_Human1_HiCedrn_cond_l2_lin.pytorch"]
    from hicdiff_amd.hicdiff_condition import GaussianDiffusion
    from hicdiff_amd.model.hicedrn_Diff import hicedrn_Diff
    d = GaussianDiffusion(hicedrn_Diff(number_resnet=2, self_condition=True), image_size=16, timesteps=1000, loss_type="l2", beta_schedule="linear")
    d.load_state_dict(torch.load(tmp_path / names[1], map_location="cpu"))          # strict


def test_train_two_ranks_stay_in_sync(tmp_path):
    """The N-rank path on the one GPU of the box (gloo, both ranks on device 0): ranks draw different batches, timesteps and
    noise, sum the flat gradient over the ranks every step (stage by stage behind the backward pass), and must hold identical parameters at the end."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HICDIFF_DEVICE="0", HICDIFF_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29731",
           os.path.join(root, "train.py"), "-u", "", "-b", "4", "-e", "2", "--resnet-blocks", "1", "--tile", "16", "--tiles-per-epoch", "18",
           "--lr", "5e-4", "--weights-dir", str(tmp_path), "--print-checksum"]        # 18 tiles = 4 full batches + a ragged fifth: an ODD batch count over 2 ranks
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    sums = {}
    for l in out.stdout.splitlines():
        if l.startswith("{") and "param_sum" in l:
            j = json.loads(l)
            sums[j["rank"]] = (j["param_sum"], j["param_abs_sum"])
    assert set(sums) == {0, 1} and sums[0] == sums[1]
    single = subprocess.run([sys.executable, os.path.join(root, "train.py"), "-u", "", "-b", "4", "-e", "2", "--resnet-blocks", "1", "--tile", "16",
                             "--tiles-per-epoch", "18", "--lr", "5e-4", "--weights-dir", str(tmp_path / "one"), "--print-checksum"],
                            env=env, capture_output=True, text=True, timeout=600)
    assert single.returncode == 0, single.stderr[-2000:]
    one = [json.loads(l) for l in single.stdout.splitlines() if "param_sum" in l][0]
    assert (one["param_sum"], one["param_abs_sum"]) != sums[0]            # the two-rank run really averaged different gradients
    # the stage-by-stage sums behind the backward pass (the default above) and one all-reduce of the whole buffer in Adam.step agree bit for bit
    whole = subprocess.run([("29733" if c == "29731" else c) for c in cmd], env=dict(env, HICDIFF_DP_OVERLAP="0"), capture_output=True, text=True, timeout=600)
    assert whole.returncode == 0, whole.stderr[-2000:]
    sums0 = {json.loads(l)["rank"]: (json.loads(l)["param_sum"], json.loads(l)["param_abs_sum"]) for l in whole.stdout.splitlines() if l.startswith("{") and "param_sum" in l}
    assert sums0 == sums


def test_train_plain_bf16_option():
    """The optional mixed-precision arithmetic (one bf16 MFMA per product in the 256-channel convolutions and in the weight-gradient
    GEMM; fp32 master weights, accumulation and everything else): gradients carry bf16 rounding -- within 3e-2 of each tensor's
    largest fp32 entry, loss within 1e-2 -- the step repeats bit for bit, and a 10-step Adam run still follows the oracle's loss curve."""
    from hicdiff_amd.optim import Adam
    from oracle import diffusion as OD, train as OTR
    B, S, nres = 3, 40, 2                                   # 40x40: the wide 8-wave convolution variant, which has the bf16 form
    d = _diffusion("cond", nres, S)
    d.model.train_precision = "bf16"
    d.train()
    sd, cfg = _oracle_sd("cond", nres)
    buf = OD.diffusion_buffers("linear", 1000)
    x0, lq = tiles(91, B, S), tiles(92, B, S)
    gen = torch.Generator().manual_seed(3)
    t, eps = torch.randint(0, 1000, (B,), generator=gen), torch.randn(x0.shape, generator=gen)
    ol, og = OTR.loss_and_grads(sd, cfg, buf, x0, t, eps, lq, "l2")
    val = d.p_losses([lq.cuda(), x0.cuda()], t.cuda(), eps.cuda())
    val.backward()
    assert d.model.__dict__["_hd_trainer"].precision == "bf16"
    assert abs(float(val.detach()) - float(ol)) <= 1e-2 * float(ol)
    errs = {k: rel_err(og[k], p.grad) for k, p in d.model.named_parameters()}
    assert max(errs.values()) <= 3e-2, max(errs.items(), key=lambda kv: kv[1])
    assert max(errs.values()) > 1e-4                        # it really is the cheaper arithmetic
    first = {k: p.grad.clone() for k, p in d.model.named_parameters()}
    for p in d.model.parameters():
        p.grad = None
    val2 = d.p_losses([lq.cuda(), x0.cuda()], t.cuda(), eps.cuda())
    val2.backward()
    assert torch.equal(val, val2) and all(torch.equal(first[k], p.grad) for k, p in d.model.named_parameters())
    opt = Adam(d.parameters(), lr=2e-4)
    opt.zero_grad()                                         # (gradients accumulate over backward() calls, as in torch: start the run clean)
    om, ov = {k: torch.zeros_like(v) for k, v in sd.items()}, {k: torch.zeros_like(v) for k, v in sd.items()}
    for step in range(1, 11):
        gen = torch.Generator().manual_seed(100 + step)
        t, eps = torch.randint(0, 1000, (B,), generator=gen), torch.randn(x0.shape, generator=gen)
        loss = d.p_losses([lq.cuda(), x0.cuda()], t.cuda(), eps.cuda())
        loss.backward()
        opt.step()
        opt.zero_grad()
        ol, og = OTR.loss_and_grads(sd, cfg, buf, x0, t, eps, lq, "l2")
        OTR.adam_step(sd, og, om, ov, step, lr=2e-4)
        assert abs(float(loss.detach()) - float(ol)) <= 3e-2 * float(ol), step


@pytest.mark.parametrize("B,S,C0,C1,Cout,KT,aff,plain", [
    (2, 10, 64, 0, 64, 3, False, 0),       # 64 input channels: the activation image is padded to 128 rows
    (3, 5, 128, 64, 128, 3, True, 0),      # channel concat + normalised-activation input, 5x5 map (row pitch 16)
    (2, 20, 128, 64, 64, 1, False, 0),     # 1x1 filter (res_conv of an up block)
    (1, 64, 64, 0, 64, 3, True, 0),
    (2, 40, 256, 0, 256, 3, False, 1),     # plain bf16 products
    (2, 8, 512, 512, 512, 3, False, 0),    # the widest UNet layer
])
def test_weight_gradient_component_any_shape(B, S, C0, C1, Cout, KT, aff, plain):
    """The weight-gradient component (operand rewrite + GEMM + reduce) that the hicedrn trainer uses at 256 x 256 channels, on
    the shapes the UNet's layers have, against torch's conv2d weight gradient (CPU fp32).  Groundwork for the UNet training step."""
    import ctypes as C
    from hicdiff_amd import _lib as L
    lib = L.load()
    lib.hd_debug_conv_wgrad.restype = C.c_int
    lib.hd_debug_conv_wgrad.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    gen = torch.Generator().manual_seed(B * 1000 + S)
    Cin = C0 + C1
    x = torch.randn((B, Cin, S, S), generator=gen)
    g = torch.randn((B, Cout, S, S), generator=gen) * 0.1
    A = torch.rand((B, Cin), generator=gen) + 0.5 if aff else None
    Bv = torch.randn((B, Cin), generator=gen) * 0.3 if aff else None
    xin = torch.nn.functional.silu(x * A[:, :, None, None] + Bv[:, :, None, None]) if aff else x
    ref = torch.nn.grad.conv2d_weight(xin, (Cout, Cin, KT, KT), g, padding=KT // 2)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    x0, x1 = nhwc(x[:, :C0]), (nhwc(x[:, C0:]) if C1 else None)
    gd, out = nhwc(g), torch.empty((Cout, Cin, KT, KT), device="cuda")
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p()
    Ad, Bd = (A.cuda().contiguous(), Bv.cuda().contiguous()) if aff else (None, None)
    rc = lib.hd_debug_conv_wgrad(ptr(x0), C0, ptr(x1), C1, ptr(gd), B, S, S, Cout, KT, ptr(Ad), ptr(Bd), plain, ptr(out),
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    assert rel_err(ref, out) <= (2e-2 if plain else 1e-4)


@pytest.mark.parametrize("B,S,C0,C1,Cout,KT,aff", [
    (2, 10, 64, 0, 64, 3, 0),          # 64-row tile; map width 10 (padded pitch 11)
    (3, 5, 128, 64, 128, 3, 0),        # channel concat, 192 input channels in a 256-row tile pair, 5x5 map
    (2, 20, 128, 64, 64, 1, 0),        # 1x1 filter
    (1, 64, 64, 0, 64, 3, 0),
    (2, 8, 512, 512, 512, 3, 0),       # the widest UNet layer
    (5, 16, 256, 0, 384, 1, 0),        # to_qkv
    (4, 32, 128, 0, 128, 3, 0),
    (3, 5, 128, 0, 128, 3, 1),         # normalised-activation input; three samples inside one 64-position slice
    (2, 64, 64, 0, 64, 3, 1),
    (2, 16, 256, 0, 256, 3, 2),        # + SR3's additive embedding after the SiLU
    (1, 31, 128, 0, 64, 3, 0),         # nine-tap kernel: 64 + 2 (W + 1) = 128 positions, the 128-slot ring exactly full
    (2, 32, 192, 0, 64, 3, 0),         # one more column: the 256-slot ring; 192 input channels = a padded second row tile
    (1, 33, 128, 64, 384, 3, 1),       # odd width, concat + affine input, six column tiles
    (1, 64, 128, 0, 64, 3, 1),         # 194 of the ring's 256 slots in use (the hicedrn shape at one sample)
])
def test_weight_gradient_direct_from_nhwc(B, S, C0, C1, Cout, KT, aff):
    """wgrad_direct_kernel (operands read as the forward left them, transposed LDS reads) against torch's conv2d weight gradient and
    the bias gradient against the pixel sum."""
    import ctypes as C
    P = C.c_void_p
    fn = _dbg("hd_debug_conv_wgrad_direct", [P, C.c_int, P, C.c_int, P] + [C.c_int] * 5 + [P, P, P, P, P, C.c_int, P])
    gen = torch.Generator().manual_seed(B * 1000 + S + KT)
    Cin = C0 + C1
    x = torch.randn((B, Cin, S, S), generator=gen)
    g = torch.randn((B, Cout, S, S), generator=gen) * 0.1
    A = torch.rand((B, Cin), generator=gen) + 0.5 if aff else None
    Bv = torch.randn((B, Cin), generator=gen) * 0.3 if aff else None
    E = torch.randn((B, Cin), generator=gen) * 0.5 if aff == 2 else None
    xin = x
    if aff:
        xin = torch.nn.functional.silu(x * A[:, :, None, None] + Bv[:, :, None, None])
        if aff == 2:
            xin = xin + E[:, :, None, None]
    ref = torch.nn.grad.conv2d_weight(xin, (Cout, Cin, KT, KT), g, padding=KT // 2)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    x0, x1 = nhwc(x[:, :C0]), (nhwc(x[:, C0:]) if C1 else None)
    gd, out, db = nhwc(g), torch.empty((Cout, Cin, KT, KT), device="cuda"), torch.empty(Cout, device="cuda")
    dev = lambda t: t.cuda().contiguous() if t is not None else None
    Ad, Bd, Ed = dev(A), dev(Bv), dev(E)
    ptr = lambda t: P(t.data_ptr()) if t is not None else P()
    rc = fn(ptr(x0), C0, ptr(x1), C1, ptr(gd), B, S, S, Cout, KT, ptr(out), ptr(db), ptr(Ad), ptr(Bd), ptr(Ed), 0, P(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    assert rel_err(ref, out) <= 1e-4
    assert rel_err(g.sum(dim=(0, 2, 3)), db) <= 1e-5


def _dbg(name, argtypes):
    import ctypes as C
    from hicdiff_amd import _lib as L
    fn = getattr(L.load(), name)
    fn.restype, fn.argtypes = C.c_int, argtypes
    return fn


@pytest.mark.parametrize("B,S,Cc,film", [(3, 8, 64, True), (2, 20, 128, True), (2, 5, 512, False), (1, 64, 64, True)])
def test_groupnorm_film_silu_backward_component(B, S, Cc, film):
    """d/dx, d/dgamma, d/dbeta, d/d(scale, shift) of silu(GroupNorm8(x) * (scale + 1) + shift) against torch autograd."""
    import ctypes as C
    P = C.c_void_p
    fn = _dbg("hd_debug_gn_silu_bwd", [P] * 5 + [C.c_int] * 5 + [P] * 4)
    gen = torch.Generator().manual_seed(Cc + S)
    x = (torch.randn((B, Cc, S, S), generator=gen) * 1.5 + 0.3).requires_grad_(True)
    gam = (torch.rand(Cc, generator=gen) + 0.5).requires_grad_(True)
    bet = (torch.randn(Cc, generator=gen) * 0.2).requires_grad_(True)
    fs = (torch.randn((B, 2 * Cc), generator=gen) * 0.3).requires_grad_(True) if film else None
    dy = torch.randn((B, Cc, S, S), generator=gen)
    h = torch.nn.functional.group_norm(x, 8, gam, bet, eps=1e-5)
    if film:
        h = h * (fs[:, :Cc, None, None] + 1) + fs[:, Cc:, None, None]
    torch.nn.functional.silu(h).backward(dy)
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().cuda()
    xd, gd = nhwc(x), nhwc(dy)
    dg, db = torch.empty(Cc, device="cuda"), torch.empty(Cc, device="cuda")
    df = torch.empty((B, 2 * Cc), device="cuda") if film else None
    ptr = lambda t: P(t.data_ptr()) if t is not None else P()
    fsd = fs.detach().cuda().contiguous() if film else None
    gamd, betd = gam.detach().cuda(), bet.detach().cuda()          # (kept alive: a temporary's block would be reused by the next one)
    rc = fn(ptr(xd), ptr(gd), ptr(gamd), ptr(betd), ptr(fsd), B, S, S, Cc, 8, ptr(dg), ptr(db), ptr(df),
            P(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    assert rel_err(x.grad.permute(0, 2, 3, 1), gd) <= 1e-4
    assert rel_err(gam.grad, dg) <= 1e-4 and rel_err(bet.grad, db) <= 1e-4
    if film:
        assert rel_err(fs.grad, df) <= 1e-4


@pytest.mark.parametrize("P_,Cc", [(200, 64), (1000, 128), (75, 512), (300, 256), (70000, 64)])
def test_channel_layernorm_backward_component(P_, Cc):
    import ctypes as C
    P = C.c_void_p
    fn = _dbg("hd_debug_ln_bwd", [P, P, P, C.c_longlong, C.c_int, P, P])
    gen = torch.Generator().manual_seed(P_)
    x = (torch.randn((P_, Cc), generator=gen) * 2 + 0.5).requires_grad_(True)
    gain = (torch.rand(Cc, generator=gen) + 0.5).requires_grad_(True)
    dy = torch.randn((P_, Cc), generator=gen)
    mean = x.mean(dim=1, keepdim=True)
    var = x.var(dim=1, unbiased=False, keepdim=True)
    ((x - mean) * (var + 1e-5).rsqrt() * gain).backward(dy)
    xd, dyd, dgain, gaind = x.detach().cuda(), dy.cuda().clone(), torch.empty(Cc, device="cuda"), gain.detach().cuda()
    rc = fn(P(xd.data_ptr()), P(dyd.data_ptr()), P(gaind.data_ptr()), P_, Cc, P(dgain.data_ptr()), P(torch.cuda.current_stream().cuda_stream))
    assert rc == 0 and rel_err(x.grad, dyd) <= 1e-4 and rel_err(gain.grad, dgain) <= 1e-4


@pytest.mark.parametrize("Cout,Cin,k", [(64, 64, 3), (128, 192, 3), (512, 1024, 3)])
def test_weight_standardisation_backward_component(Cout, Cin, k):
    import ctypes as C
    P = C.c_void_p
    fn = _dbg("hd_debug_ws_bwd", [P, P, C.c_int, C.c_int, P, P])
    gen = torch.Generator().manual_seed(Cout + Cin)
    w = (torch.randn((Cout, Cin, k, k), generator=gen) * 0.05).requires_grad_(True)
    dwh = torch.randn((Cout, Cin, k, k), generator=gen)
    mean = w.mean(dim=(1, 2, 3), keepdim=True)
    var = w.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
    ((w - mean) * (var + 1e-5).rsqrt()).backward(dwh)                       # src/hicdiff.py:89-97
    wd, dd, out = w.detach().cuda(), dwh.cuda(), torch.empty_like(dwh, device="cuda")
    rc = fn(P(wd.data_ptr()), P(dd.data_ptr()), Cout, Cin * k * k, P(out.data_ptr()), P(torch.cuda.current_stream().cuda_stream))
    assert rc == 0 and rel_err(w.grad, out) <= 1e-5


@pytest.mark.parametrize("B,n", [(3, 64), (2, 25)])
def test_full_attention_backward_component(B, n):
    """d(q, k, v) of the mid block's softmax attention (src/hicdiff.py:239-251) against torch autograd."""
    import ctypes as C
    P = C.c_void_p
    fn = _dbg("hd_debug_attn_full_bwd", [P, P, C.c_int, C.c_int, C.c_int, P, P])
    heads, D = 4, 32
    gen = torch.Generator().manual_seed(n)
    qkv = torch.randn((B, n, 3 * heads * D), generator=gen).requires_grad_(True)
    dout = torch.randn((B, n, heads * D), generator=gen)
    q, k, v = (t.reshape(B, n, heads, D).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=2))        # b h n d
    sim = torch.einsum("bhid,bhjd->bhij", q * D ** -0.5, k)
    out = torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), v).permute(0, 2, 1, 3).reshape(B, n, heads * D)
    out.backward(dout)
    qd, dd, res = qkv.detach().cuda(), dout.cuda(), torch.empty((B, n, 3 * heads * D), device="cuda")
    rc = fn(P(qd.data_ptr()), P(dd.data_ptr()), B, n, heads, P(res.data_ptr()), P(torch.cuda.current_stream().cuda_stream))
    assert rc == 0 and rel_err(qkv.grad, res) <= 1e-4


@pytest.mark.parametrize("B,n", [(2, 100), (3, 256), (1, 1600), (2, 4096)])
def test_linear_attention_backward_component(B, n):
    """d(q, k, v) of the LinearAttention core (src/hicdiff.py:212-224) against torch autograd."""
    import ctypes as C
    P = C.c_void_p
    fn = _dbg("hd_debug_linattn_bwd", [P, P, C.c_int, C.c_int, C.c_int, P, P])
    heads, D = 4, 32
    gen = torch.Generator().manual_seed(n + B)
    qkv = (torch.randn((B, n, 3 * heads * D), generator=gen) * 1.5).requires_grad_(True)
    dout = torch.randn((B, n, heads * D), generator=gen)
    q, k, v = (t.reshape(B, n, heads, D).permute(0, 2, 3, 1) for t in qkv.chunk(3, dim=2))        # b h d n
    q = q.softmax(dim=-2) * D ** -0.5
    k = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v / n)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q).permute(0, 3, 1, 2).reshape(B, n, heads * D)
    out.backward(dout)
    qd, dd, res = qkv.detach().cuda(), dout.cuda(), torch.empty((B, n, 3 * heads * D), device="cuda")
    rc = fn(P(qd.data_ptr()), P(dd.data_ptr()), B, n, heads, P(res.data_ptr()), P(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    for name, sl in (("q", slice(0, 128)), ("k", slice(128, 256)), ("v", slice(256, 384))):
        assert rel_err(qkv.grad[..., sl], res[..., sl]) <= 1e-4, name


def test_resampling_layers_backward_components():
    """Upsample (nearest x2 -> conv 3x3) and Downsample (pixel-unshuffle -> conv 1x1), src/hicdiff.py:72-82: the weight gradient with
    the resampling done as source addressing of the operand rewrite, and the routing of the data gradient back through it."""
    import ctypes as C
    from einops import rearrange
    P = C.c_void_p
    wg = _dbg("hd_debug_conv_wgrad", [P, C.c_int, P, C.c_int, P] + [C.c_int] * 5 + [P, P, C.c_int, P, P])
    wgd = _dbg("hd_debug_conv_wgrad_direct", [P, C.c_int, P, C.c_int, P] + [C.c_int] * 5 + [P, P, P, P, P, C.c_int, P])
    rs = _dbg("hd_debug_resample_bwd", [P] + [C.c_int] * 5 + [P, P])
    st = P(torch.cuda.current_stream().cuda_stream)
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().cuda()
    gen = torch.Generator().manual_seed(17)
    B, Cc, Cout, H = 2, 128, 64, 20                      # the map after the layer is H x H
    # Upsample: x is H/2 x H/2
    x = torch.randn((B, Cc, H // 2, H // 2), generator=gen).requires_grad_(True)
    g = torch.randn((B, Cout, H, H), generator=gen) * 0.1
    up = torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest")
    ref = torch.nn.grad.conv2d_weight(up, (Cout, Cc, 3, 3), g, padding=1)
    xd, gd, out = nhwc(x), nhwc(g), torch.empty((Cout, Cc, 3, 3), device="cuda")
    assert wg(P(xd.data_ptr()), Cc, P(), 0, P(gd.data_ptr()), B, H, H, Cout, 3, P(), P(), 2, P(out.data_ptr()), st) == 0
    assert rel_err(ref, out) <= 1e-4
    out.zero_()                                                  # the same through the direct kernel's source addressing
    assert wgd(P(xd.data_ptr()), Cc, P(), 0, P(gd.data_ptr()), B, H, H, Cout, 3, P(out.data_ptr()), P(), P(), P(), P(), 1, st) == 0
    assert rel_err(ref, out) <= 1e-4
    gup = torch.randn((B, Cc, H, H), generator=gen)
    up.backward(gup)
    gud, dx = nhwc(gup), torch.empty((B, H // 2, H // 2, Cc), device="cuda")
    assert rs(P(gud.data_ptr()), B, H // 2, H // 2, Cc, 1, P(dx.data_ptr()), st) == 0
    assert rel_err(x.grad.permute(0, 2, 3, 1), dx) <= 1e-6
    # Downsample: x is 2H x 2H with Cs channels, the 1x1 conv sees 4 Cs channels at H x H
    Cs = 64
    x = torch.randn((B, Cs, 2 * H, 2 * H), generator=gen).requires_grad_(True)
    un = rearrange(x, "b c (h p1) (w p2) -> b (c p1 p2) h w", p1=2, p2=2)
    ref = torch.nn.grad.conv2d_weight(un, (Cout, 4 * Cs, 1, 1), g)
    xd, out = nhwc(x), torch.empty((Cout, 4 * Cs, 1, 1), device="cuda")
    assert wg(P(xd.data_ptr()), 4 * Cs, P(), 0, P(gd.data_ptr()), B, H, H, Cout, 1, P(), P(), 4, P(out.data_ptr()), st) == 0
    assert rel_err(ref, out) <= 1e-4
    out.zero_()
    assert wgd(P(xd.data_ptr()), 4 * Cs, P(), 0, P(gd.data_ptr()), B, H, H, Cout, 1, P(out.data_ptr()), P(), P(), P(), P(), 2, st) == 0
    assert rel_err(ref, out) <= 1e-4
    gun = torch.randn((B, 4 * Cs, H, H), generator=gen)
    un.backward(gun)
    gnd, dx = nhwc(gun), torch.empty((B, 2 * H, 2 * H, Cs), device="cuda")
    assert rs(P(gnd.data_ptr()), B, H, H, Cs, 2, P(dx.data_ptr()), st) == 0
    assert torch.equal(x.grad.permute(0, 2, 3, 1).contiguous(), dx.cpu())


def test_first_and_last_convolution_backward_components():
    """init_conv (7x7, 1 or 2 single-channel inputs -> 64, src/hicdiff.py:279) weight gradient; final_conv (1x1, 64 -> 1, :319) both gradients."""
    import ctypes as C
    P = C.c_void_p
    fc = _dbg("hd_debug_first_conv_wgrad", [P, P, P] + [C.c_int] * 5 + [P, P])
    rd = _dbg("hd_debug_rowdot_bwd", [P, P, P, C.c_longlong, C.c_int, P, P, P])
    st = P(torch.cuda.current_stream().cuda_stream)
    gen = torch.Generator().manual_seed(23)
    for J, S, KS in ((2, 40, 7), (1, 16, 7), (2, 24, 3)):
        B, Cc = 3, 64
        x = torch.randn((B, J, S, S), generator=gen)
        g = torch.randn((B, Cc, S, S), generator=gen) * 0.1
        ref = torch.nn.grad.conv2d_weight(x, (Cc, J, KS, KS), g, padding=KS // 2)
        planes = [x[:, j].contiguous().cuda() for j in range(J)]
        gd, out = g.permute(0, 2, 3, 1).contiguous().cuda(), torch.empty((Cc, J, KS, KS), device="cuda")
        rc = fc(P(gd.data_ptr()), P(planes[0].data_ptr()), P(planes[1].data_ptr()) if J == 2 else P(), J, B, S, Cc, KS, P(out.data_ptr()), st)
        assert rc == 0 and rel_err(ref, out) <= 1e-5, (J, S, KS)
    Pn, Cc = 3 * 40 * 40, 64
    x = torch.randn((Pn, Cc), generator=gen).requires_grad_(True)
    w = torch.randn(Cc, generator=gen).requires_grad_(True)
    dout = torch.randn(Pn, generator=gen)
    (x @ w).backward(dout)
    xd, dd, wd = x.detach().cuda(), dout.cuda(), w.detach().cuda()
    dx, dw = torch.empty((Pn, Cc), device="cuda"), torch.empty(Cc, device="cuda")
    assert rd(P(xd.data_ptr()), P(dd.data_ptr()), P(wd.data_ptr()), Pn, Cc, P(dx.data_ptr()), P(dw.data_ptr()), st) == 0
    assert rel_err(x.grad, dx) <= 1e-6 and rel_err(w.grad, dw) <= 1e-5


def _unet_diffusion(kind, dim, mults, S, loss="l2"):
    from _util import product_unet
    m = product_unet(kind, dim=dim, mults=mults)
    if kind == "cond":
        from hicdiff_amd.hicdiff_condition import GaussianDiffusion
    elif kind == "sr3":
        from hicdiff_amd.hicdiff_sr3 import GaussianDiffusion
    else:
        from hicdiff_amd.hicdiff import GaussianDiffusion
    return GaussianDiffusion(m, image_size=S, timesteps=2000 if kind == "sr3" else 1000, loss_type=loss, beta_schedule="linear").cuda()


@pytest.mark.parametrize("kind,dim,mults,B,S", [("cond", 64, (1, 2), 2, 16), ("uncond", 64, (1, 2, 4), 2, 32), ("sr3", 64, (1, 2), 3, 16),
                                                ("cond", 64, (1, 2, 4, 8), 2, 40)])   # the full UNet on 40x40 tiles: maps of 40, 20, 10 and 5 pixels
def test_unet_train_gradients_vs_autograd_oracle(kind, dim, mults, B, S):
    """The UNet's native training step: every entry of every gradient against torch autograd over the oracle net (CPU fp32)."""
    from oracle import diffusion as OD, nets as ON, train as OTR, weights as W
    d = _unet_diffusion(kind, dim, mults, S)
    d.train()
    cfg = ON.UnetCfg(dim=dim, dim_mults=tuple(mults), self_condition=(kind != "uncond"), sr3=(kind == "sr3"))
    sd = W.fill_state_dict(W.unet_shapes(dim=dim, dim_mults=tuple(mults), self_condition=cfg.self_condition, sr3=cfg.sr3))
    x0, lq = tiles(51, B, S), tiles(52, B, S)
    gen = torch.Generator().manual_seed(7)
    t = torch.rand((B,), generator=gen) * 0.9 + 0.05 if kind == "sr3" else torch.randint(0, 1000, (B,), generator=gen)
    eps = torch.randn(x0.shape, generator=gen)
    ol, og = OTR.loss_and_grads(sd, cfg, OD.diffusion_buffers("linear", 1000), x0, t, eps, None if kind == "uncond" else lq, "l2")
    val = _loss(d, kind, lq.cuda(), x0.cuda(), t.cuda(), eps.cuda())
    assert val.requires_grad
    val.backward()
    assert abs(float(val.detach()) - float(ol)) <= 1e-4 * float(ol)
    errs = {k: rel_err(og[k], p.grad) for k, p in d.model.named_parameters()}
    bad = {k: v for k, v in errs.items() if not v <= 1e-3}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:12]


def test_unet_train_plain_bf16_option():
    """The UNet's step under the optional bf16 arithmetic (one bf16 MFMA per product in every 32-channel-slice convolution -- forward and data
    gradient -- and in the weight-gradient GEMMs; fp32 master weights, accumulation, normalisations and attention): loss within 1e-2, every
    gradient tensor within 5e-2 of its largest fp32 entry, and the step repeats bit for bit."""
    from oracle import diffusion as OD, nets as ON, train as OTR, weights as W
    kind, dim, mults, B, S = "cond", 64, (1, 2, 4, 8), 2, 40
    d = _unet_diffusion(kind, dim, mults, S)
    d.model.train_precision = "bf16"
    d.train()
    cfg = ON.UnetCfg(dim=dim, dim_mults=tuple(mults), self_condition=True, sr3=False)
    sd = W.fill_state_dict(W.unet_shapes(dim=dim, dim_mults=tuple(mults), self_condition=True, sr3=False))
    x0, lq = tiles(51, B, S), tiles(52, B, S)
    gen = torch.Generator().manual_seed(7)
    t, eps = torch.randint(0, 1000, (B,), generator=gen), torch.randn(x0.shape, generator=gen)
    ol, og = OTR.loss_and_grads(sd, cfg, OD.diffusion_buffers("linear", 1000), x0, t, eps, lq, "l2")
    val = _loss(d, kind, lq.cuda(), x0.cuda(), t.cuda(), eps.cuda())
    val.backward()
    assert d.model.__dict__["_hd_trainer"].precision == "bf16"
    assert abs(float(val.detach()) - float(ol)) <= 1e-2 * float(ol)
    errs = {k: rel_err(og[k], p.grad) for k, p in d.model.named_parameters()}
    print(f"unet bf16 option: loss {float(val.detach()):.6f} vs {float(ol):.6f}; gradient errors max {max(errs.values()):.2e}, median {sorted(errs.values())[len(errs) // 2]:.2e}")
    assert max(errs.values()) <= 5e-2, max(errs.items(), key=lambda kv: kv[1])
    assert max(errs.values()) > 1e-3                        # it really is the cheaper arithmetic
    first = {k: p.grad.clone() for k, p in d.model.named_parameters()}
    for p in d.model.parameters():
        p.grad = None
    val2 = _loss(d, kind, lq.cuda(), x0.cuda(), t.cuda(), eps.cuda())
    val2.backward()
    assert torch.equal(val, val2) and all(torch.equal(first[k], p.grad) for k, p in d.model.named_parameters())


@pytest.mark.parametrize("kind", ["cond", "uncond", "sr3"])
def test_unet_train_two_steps_golden(kind):
    """Two steps of loss.backward() + Adam(lr=2e-5) on the two-level UNet the reference ran (make_golden.py::case_train_unet)."""
    from hicdiff_amd.optim import Adam
    from oracle.train import sample_of
    g = golden("train_unet")
    d = _unet_diffusion(kind, 64, (1, 2), 16)
    d.train()
    opt = Adam(d.parameters(), lr=2e-5)
    x0, lq = g["x0"].cuda(), g["lq"].cuda()
    for step in (1, 2):
        t, eps = g[f"{kind}_s{step}_t"].cuda(), g[f"{kind}_s{step}_eps"].cuda()
        loss = _loss(d, kind, lq, x0, t, eps)
        loss.backward()
        assert abs(float(loss.detach()) - float(g[f"{kind}_s{step}_loss"])) <= 1e-4 * float(g[f"{kind}_s{step}_loss"])
        for k, p in d.model.named_parameters():
            ref_s, ref_n = g[f"{kind}_s{step}_grad_sample/{k}"], float(g[f"{kind}_s{step}_grad_norm/{k}"])
            got = p.grad.detach().cpu()
            assert abs(float(got.norm()) - ref_n) <= 1e-3 * ref_n + 1e-12, (step, k)
            scale = max(float(ref_s.abs().max()), ref_n / got.numel() ** 0.5)
            assert float((sample_of(got) - ref_s).abs().max()) <= 1e-3 * scale, (step, k)
        opt.step()
        opt.zero_grad()
        for k, p in d.model.named_parameters():
            ref = g[f"{kind}_s{step}_param_sample/{k}"]
            # Adam moves a parameter by ~lr per step whatever the size of its gradient, so where the gradient is tiny against its tensor's
            # largest entry an error of 1e-3 of that entry can turn the update around: compare where the gradient is not tiny
            gs = g[f"{kind}_s{step}_grad_sample/{k}"].abs()
            keep = gs > 0.05 * gs.max()
            diff = (sample_of(p.detach().cpu()) - ref).abs()
            assert float(diff[keep].max()) <= 5e-2 * 2e-5 * step + 4e-7 * float(ref.abs().max()), (step, k)
            assert float(diff.max()) <= 2.1 * 2e-5 * step, (step, k)                 # and nowhere by more than a turned-around update
    assert list(d.model.state_dict().keys()) == [k for k, _ in d.model.named_parameters()]


def test_unet_train_full_network_properties_and_loop():
    """The full UNet (dim 64, dim_mults (1, 2, 4, 8)) at 64x64: the step repeats bit for bit, a batch's gradient is the mean of its halves',
    and a few Adam steps on a fixed batch lower the loss; eval-mode sampling afterwards uses the updated weights."""
    from hicdiff_amd.optim import Adam
    d = _unet_diffusion("cond", 64, (1, 2, 4, 8), 64)
    d.train()
    B = 4
    x0, lq = tiles(61, B, 64).cuda(), tiles(62, B, 64).cuda()
    gen = torch.Generator().manual_seed(11)
    t, eps = torch.randint(0, 1000, (B,), generator=gen).cuda(), torch.randn(x0.shape, generator=gen).cuda()

    def grads(sl):
        for p in d.model.parameters():
            p.grad = None
        loss = d.p_losses([lq[sl], x0[sl]], t[sl], eps[sl])
        loss.backward()
        return float(loss.detach()), {k: p.grad.clone() for k, p in d.model.named_parameters()}

    l_all, g_all = grads(slice(0, B))
    l_again, g_again = grads(slice(0, B))
    assert l_all == l_again and all(torch.equal(g_all[k], g_again[k]) for k in g_all)
    l_a, g_a = grads(slice(0, B // 2))
    l_b, g_b = grads(slice(B // 2, B))
    assert abs(l_all - 0.5 * (l_a + l_b)) <= 1e-5 * l_all
    worst = max(rel_err(g_all[k], 0.5 * (g_a[k] + g_b[k])) for k in g_all)
    assert worst <= 5e-4, worst
    opt = Adam(d.parameters(), lr=2e-4)
    before = d.model(x0, t, lq).clone()
    losses = []
    for _ in range(6):
        loss = d.p_losses([lq, x0], t, eps)
        loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.9 * losses[0], losses
    d.eval()
    with torch.no_grad():
        after = d.model(x0, t, lq)
    assert rel_err(before, after) > 1e-3


@pytest.mark.parametrize("net", ["hicedrn", "unet", "unet_small"])
def test_gradient_stages_are_final_at_their_event(net):
    """Gradient stages (include/hicdiff_hip.h): every parameter slot belongs to one stage; the stages' events are recorded in stage order;
    and -- what an all-reduce starting at the event relies on -- no kernel writes a slot after its stage's event.  The trainer's debug
    snapshot copies each stage's slots in stream order right behind its event: the snapshot must equal the gradients the step ends with,
    bit for bit, and those must equal the gradients of the same step taken without any of this (the golden / oracle tests run that way)."""
    import ctypes as C
    from hicdiff_amd import _lib as L
    from hicdiff_amd._training import trainer_for
    if net == "hicedrn":
        d, B, S = _diffusion("cond", 6, 16), 3, 16                      # six blocks: four block stages of 2, 2, 1, 1 + the final stage
    elif net == "unet":
        d, B, S = _unet_diffusion("cond", 64, (1, 2, 4, 8), 32), 2, 32
    else:
        d, B, S = _unet_diffusion("uncond", 64, (1, 2), 16), 2, 16
    d.train()
    x0, lq = tiles(71, B, S).cuda(), tiles(72, B, S).cuda()
    gen = torch.Generator().manual_seed(5)
    t, eps = torch.randint(0, 1000, (B,), generator=gen).cuda(), torch.randn(x0.shape, generator=gen).cuda()
    cond = net != "unet_small"
    step = lambda: d.p_losses([lq, x0], t, eps) if cond else d.p_losses(x0, t, eps)
    step().backward()
    tr = trainer_for(d.model, B, S)
    plain = tr.grads.clone()
    lib = L.load()
    nst = lib.hd_train_stage_count(tr.h)
    assert 2 <= nst <= 5 and sorted(set(tr.slot_stage)) == list(range(nst))      # no empty stage, every slot mapped
    names = [s[0] for s in tr.slots]
    last = [n for n, k in zip(names, tr.slot_stage) if k == nst - 1]
    first = [n for n, k in zip(names, tr.slot_stage) if k == 0]
    if net == "hicedrn":
        assert "tail.weight" in first and "body_tail.weight" in first and "head.weight" in last
        assert len(last) == 6                                                                    # head (2) + time MLP (4): everything else is final earlier
        by_block = {}
        for n, k in zip(names, tr.slot_stage):
            if n.startswith("body."):
                by_block.setdefault(n.split(".")[1], set()).add(k)
        assert len(by_block) == 6 and all(len(v) == 1 for v in by_block.values())                # a block's convolution and FiLM projection: one stage
        assert [min(by_block[str(i)]) for i in range(6)] == [3, 2, 2, 1, 0, 0]                   # back to front
    else:
        assert "final_conv.weight" in first and "init_conv.weight" in last and "time_mlp.1.weight" in last
    fn = lib.hd_debug_train_stage_snapshot
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
    snap = torch.full_like(tr.grads, float("nan"))
    assert fn(tr.h, C.c_void_p(snap.data_ptr())) == 0
    try:
        for p in d.model.parameters():
            p.grad = None                          # (with live .grad views the step would ADD to them: gradient accumulation)
        tr.grads.fill_(float("nan"))
        step().backward()
        torch.cuda.synchronize()
    finally:
        fn(tr.h, None)
    used = torch.zeros_like(snap, dtype=torch.bool)          # the flat buffer aligns its slots: the gaps between them belong to nobody
    for (name, off, shape), k in zip(tr.slots, tr.slot_stage):
        used[off:off + int(np.prod(shape))] = True
    assert not torch.isnan(tr.grads[used]).any()
    assert not torch.isnan(snap[used]).any(), "a slot was in no stage's snapshot"
    assert torch.equal(snap[used], tr.grads[used]), "a gradient slot was written after its stage's event"
    assert torch.equal(tr.grads[used], plain[used])
    # a side stream that waits for stage 0 only may read stage 0's slots: the wait itself must succeed and order after the event
    side = torch.cuda.Stream()
    assert lib.hd_train_stage_wait(tr.h, 0, C.c_void_p(side.cuda_stream)) == 0
    assert lib.hd_train_stage_wait(tr.h, nst, C.c_void_p(side.cuda_stream)) == L.HD_EINVAL
    side.synchronize()


def test_staged_reduction_starts_before_the_backward_pass_ends(monkeypatch):
    """The point of the gradient stages: stage k's bucket is packed and its collective queued while the kernels of the later stages still run.
    One process, the collective replaced by a recorder that drops a timing event on the side stream where the all-reduce would start:
    every stage but the last must be ready to travel BEFORE the step's last kernel finishes, in stage order, and the bucket handed to
    the collective must already hold the stage's final gradients (compared after the step with the flat buffer)."""
    from hicdiff_amd._training import StagedReducer, trainer_for
    d = _diffusion("cond", 8, 64)
    d.train()
    B, S = 16, 64                 # 14 ms of kernels against ~1-6 ms of host enqueue: at 4 tiles the step is host-bound and nothing can overlap
    x0, lq = tiles(91, B, S).cuda(), tiles(92, B, S).cuda()
    gen = torch.Generator().manual_seed(3)
    t, eps = torch.randint(0, 1000, (B,), generator=gen).cuda(), torch.randn(x0.shape, generator=gen).cuda()
    d.p_losses([lq, x0], t, eps).backward()                 # creates the trainer, warms every kernel up
    tr = trainer_for(d.model, B, S)
    red = StagedReducer(tr.grads, tr.slots, tr.slot_stage)
    marks, seen = [], []

    class Work:
        def wait(self):
            return True

    def fake_all_reduce(bucket, group=None, async_op=False):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())              # the side stream, behind the stage's event wait and the pack
        marks.append(ev)
        seen.append(bucket.clone())                          # (also on the side stream: what the collective would send)
        return Work()

    monkeypatch.setattr(torch.distributed, "all_reduce", fake_all_reduce)
    for p in d.model.parameters():
        p.grad = None
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    loss = d.p_losses([lq, x0], t, eps)                     # queues the whole step on the current stream
    red.launch(tr._wait_stage)                               # ... and the per-stage packs + collectives behind their events
    end.record()                                             # behind the step's last kernel on the compute stream
    red.finish()
    torch.cuda.synchronize()
    nst = len(red.runs)
    assert len(marks) == nst == 5
    total = start.elapsed_time(end)
    at = [start.elapsed_time(m) for m in marks]
    assert all(a < b for a, b in zip(at, at[1:])), at                      # stage order
    assert all(a < total for a in at[:-1]), (at, total)                     # ready to travel while later stages are still being computed
    assert at[0] < 0.8 * total, (at, total)                                 # stage 0 (the output end) well before the end of the step (measured: 0.57)
    for k in range(nst):                                                    # what was handed over is final
        pos = 0
        for off, n in red.runs[k]:
            assert torch.equal(seen[k][pos:pos + n], tr.grads[off:off + n]), k
            pos += n
    loss.backward()


def test_adam_state_dict_round_trip_and_torch_layout():
    """The per-epoch checkpoint form of the pretrain scripts (pretrain/train_hicedrn_Diff.py:93-96: {'epoch', 'model_state_dict',
    'optimizer_state_dict'}): Adam.state_dict() has torch.optim.Adam's layout (it loads into one), and a run resumed from
    (model_state_dict, optimizer_state_dict) after two steps takes the same third step, bit for bit, as the run that never stopped."""
    import copy
    import io
    from hicdiff_amd.optim import Adam
    B, S = 2, 16
    x0, lq = tiles(81, B, S).cuda(), tiles(82, B, S).cuda()
    gen = torch.Generator().manual_seed(9)
    draws = [(torch.randint(0, 1000, (B,), generator=gen).cuda(), torch.randn(x0.shape, generator=gen).cuda()) for _ in range(3)]

    def run(d, opt, steps):
        for t, eps in steps:
            d.p_losses([lq, x0], t, eps).backward()
            opt.step()
            opt.zero_grad()

    d = _diffusion("cond", 2, S)
    d.train()
    opt = Adam(d.parameters(), lr=1e-3)
    assert opt.state_dict()["state"] == {}
    run(d, opt, draws[:2])
    buf = io.BytesIO()
    torch.save({"epoch": 2, "model_state_dict": d.state_dict(), "optimizer_state_dict": opt.state_dict()}, buf)     # the reference's checkpoint form
    run(d, opt, draws[2:])
    want = {k: v.detach().clone() for k, v in d.model.named_parameters()}

    buf.seek(0)
    ck = torch.load(buf, map_location="cuda")
    sd = ck["optimizer_state_dict"]
    n = len(list(d.parameters()))
    assert sorted(sd["state"]) == list(range(n)) and sd["param_groups"][0]["params"] == list(range(n))
    assert all(float(st["step"]) == 2.0 and st["exp_avg"].shape == p.shape for st, p in zip(sd["state"].values(), d.parameters()))
    ref_opt = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in d.parameters()], lr=1e-3)
    ref_opt.load_state_dict(copy.deepcopy(sd))                         # torch accepts the layout
    assert torch.equal(ref_opt.state[ref_opt.param_groups[0]["params"][3]]["exp_avg_sq"], sd["state"][3]["exp_avg_sq"])

    d2 = _diffusion("cond", 2, S)
    d2.load_state_dict(ck["model_state_dict"])
    d2.train()
    opt2 = Adam(d2.parameters(), lr=5e-2)                              # lr comes back from the checkpoint
    opt2.load_state_dict(sd)
    assert opt2.param_groups[0]["lr"] == 1e-3
    assert torch.equal(opt2.state_dict()["state"][5]["exp_avg"], sd["state"][5]["exp_avg"])      # readable before the first step, too
    run(d2, opt2, draws[2:])
    for k, v in d2.model.named_parameters():
        assert torch.equal(v, want[k]), k
    assert float(opt2.state_dict()["state"][0]["step"]) == 3.0


def test_train_cli_unet(tmp_path, capsys):
    """train.py --arch unet: the full UNet (64, (1,2,4,8)) trains natively on 32x32 tiles; loss falls; checkpoints carry the Unet tag."""
    import json
    import train
    train.main(["-u", "", "-b", "4", "-e", "3", "--arch", "unet", "--tile", "32", "--tiles-per-epoch", "16", "--lr", "3e-4", "--weights-dir", str(tmp_path)])
    lines = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 3 and lines[-1]["train/loss"] < 0.9 * lines[0]["train/loss"], lines
    assert sorted(p.name for p in tmp_path.iterdir()) == ["bestg_40000_c32_s32_Human1_unet_cond_l2_lin.pytorch", "finalg_40000_c32_s32_Human1_unet_cond_l2_lin.pytorch"]
