"""Host-side logic that needs no GPU: the C-ABI library loads and exports every declared symbol,
the mirror classes reproduce the reference's state-dict layout and schedule buffers, the product
path refuses to run without a device, and the tile sharding / all-gather logic (world_size-2 gloo)."""
import json
import os
import re
import socket
import subprocess
import sys
import time

import pytest
import torch

from _util import GOLDEN, golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for hdr in ("hicdiff_hip.h", "hicdiff_hip_debug.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(hd_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_loads_and_exports_every_declared_symbol():
    import ctypes
    from hicdiff_amd import _lib
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/*.h but not exported"
    assert set(_lib.SYMBOLS) <= declared
    assert b"gfx950" in ctypes.c_char_p(lib.hd_version()).value


def test_create_validates_architecture_without_a_gpu_call():
    import ctypes as C
    from hicdiff_amd import _lib
    lib = _lib.load()
    bad = _lib.HdArchDesc()
    bad.kind, bad.dim, bad.n_mults, bad.channels, bad.groups = _lib.HD_ARCH_UNET, 20, 2, 1, 8
    ctx = C.c_void_p()
    assert lib.hd_create(C.byref(ctx), 0, C.byref(bad)) == _lib.HD_EINVAL   # dim % 16
    assert b"multiple of 16" in lib.hd_last_error(None)
    bad.dim, bad.channels = 64, 3
    assert lib.hd_create(C.byref(ctx), 0, C.byref(bad)) == _lib.HD_EINVAL


def test_missing_library_is_a_loud_failure(tmp_path, monkeypatch):
    from hicdiff_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        _lib.load()


def test_state_dict_layout_matches_reference_checkpoints():
    import hicdiff_amd.hicdiff as h0
    import hicdiff_amd.hicdiff_condition as h1
    import hicdiff_amd.hicdiff_sr3 as h2
    from hicdiff_amd.model.hicedrn_Diff import hicedrn_Diff
    from hicdiff_amd.model.hicedrn_sr3_Diff import hicedrn_Diff as hicedrn_sr3
    inv = json.load(open(os.path.join(GOLDEN, "param_inventory.json")))
    cases = {
        "unet_uncond": h0.Unet(64), "unet_cond": h1.Unet(64), "unet_sr3": h2.Unet(64, noise_level_emb=True),
        "hicedrn_uncond": hicedrn_Diff(), "hicedrn_cond": hicedrn_Diff(self_condition=True),
        "hicedrn_sr3": hicedrn_sr3(self_condition=True),
    }
    for name, m in cases.items():
        assert [(k, list(v.shape)) for k, v in m.state_dict().items()] == [tuple(x) for x in map(tuple, inv[name])], name
    d = h0.GaussianDiffusion(h0.Unet(16, dim_mults=(1, 2)), image_size=16, timesteps=10, beta_schedule="linear")
    keys = list(d.state_dict())
    assert [k for k in keys if not k.startswith("model.")] == [k for k, _ in inv["diffusion_buffers"]]
    assert sum(k.startswith("model.") for k in keys) == len(h0.Unet(16, dim_mults=(1, 2)).state_dict())
    # a reference checkpoint (flat, 'model.'-prefixed keys) loads strictly
    d2 = h0.GaussianDiffusion(h0.Unet(16, dim_mults=(1, 2)), image_size=16, timesteps=10, beta_schedule="linear")
    d2.load_state_dict(d.state_dict(), strict=True)


@pytest.mark.parametrize("sched,T", [("linear", 50), ("linear", 1000), ("sigmoid", 2000), ("cosine", 1000)])
def test_schedule_buffers_bit_exact_in_product(sched, T):
    import hicdiff_amd.hicdiff as h0
    g = golden("schedules")
    d = h0.GaussianDiffusion(h0.Unet(16, dim_mults=(1, 2)), image_size=16, timesteps=T, beta_schedule=sched)
    for name, buf in d.named_buffers():
        assert torch.equal(g[f"{sched}_{T}_{name}"], buf), name
    import hicdiff_amd.hicdiff_sr3 as h2
    d = h2.GaussianDiffusion(h2.Unet(16, dim_mults=(1, 2), noise_level_emb=True), image_size=16, timesteps=2000)
    assert torch.equal(d.sqrt_alphas_cumprod_prev, g["linear_2000_sqrt_alphas_cumprod_prev"])


def test_reference_guards_are_kept():
    import hicdiff_amd.hicdiff as h0
    net = h0.Unet(16, dim_mults=(1, 2))
    with pytest.raises(ValueError):
        h0.GaussianDiffusion(net, image_size=16, beta_schedule="quadratic")
    with pytest.raises(AssertionError):
        h0.GaussianDiffusion(net, image_size=16, objective="pred_eps")
    with pytest.raises(AssertionError):
        h0.GaussianDiffusion(net, image_size=16, timesteps=10, sampling_timesteps=20, beta_schedule="linear")
    with pytest.raises(NotImplementedError):
        h0.Unet(16, learned_sinusoidal_cond=True)
    d = h0.GaussianDiffusion(net, image_size=16, timesteps=100, sampling_timesteps=10, beta_schedule="linear")
    assert d.is_ddim_sampling and d.num_timesteps == 100


def test_product_path_refuses_cpu_tensors():
    import hicdiff_amd.hicdiff as h0
    net = h0.Unet(16, dim_mults=(1, 2))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 1, 16, 16), torch.zeros(1, dtype=torch.long))
    from hicdiff_amd.functions.H_func import MakeFunc
    sr = MakeFunc("sr4", 1, 16, device="cpu")              # the operator's tables are built on the host; applying it needs the GPU
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sr.Vt(torch.zeros(1, 1, 16, 16))
    with pytest.raises(ValueError):
        MakeFunc("no_such_degradation", 1, 16)
    H = MakeFunc("deno", 1, 4)
    v = torch.arange(32.).reshape(2, 1, 4, 4)
    assert torch.equal(H.H(v), v.reshape(2, -1)) and torch.equal(H.H_pinv(v), v.reshape(2, -1))


def test_shard_ranges_cover_all_tiles():
    from hicdiff_amd.sharding import shard_range
    for n in (0, 1, 7, 8, 255, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from hicdiff_amd.sharding import sample_sharded, all_gather_tiles
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
n, S = 7, 4
def run_local(start, count):   # stand-in for a per-rank sampling loop keyed by the global tile index
    idx = torch.arange(start, start + count, dtype=torch.float32).view(-1, 1, 1, 1)
    return idx * torch.ones(count, 1, S, S) + 0.5
full = sample_sharded(run_local, n, dist)
want = run_local(0, n)
assert torch.equal(full, want), (full[:, 0, 0, 0], want[:, 0, 0, 0])
even = all_gather_tiles(run_local(dist.get_rank() * 3, 3), dist)
assert torch.equal(even, run_local(0, 3 * dist.get_world_size()))
t = torch.tensor([float(dist.get_rank() + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)      # bench.py's max-over-ranks timing reduction
assert t.item() == dist.get_world_size()
dist.destroy_process_group()
print("ok")
"""


def test_sharded_sampling_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out.decode()
        assert b"ok" in out


_TRAIN_WORKER = r"""
import os, sys, types, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
import hicdiff_amd.optim as O

calls = []
class FakeLib:                                   # the HIP library is not needed for the collective logic: record what Adam.step would launch
    def hd_adam_step(self, p, g, m, v, n, lr, b1, b2, eps, step, scale, st):
        calls.append((n, lr, step, scale))
        return 0
O.L.load = lambda: FakeLib()
class FakeStream:
    cuda_stream = 0
class FakeCtx:
    def __enter__(self): return self
    def __exit__(self, *a): return False
torch.cuda.device = lambda d: FakeCtx()
torch.cuda.current_stream = lambda: FakeStream()

class FakeTrainer:                               # what optim.Adam touches of a NativeTrainer
    def __init__(self, model):
        self.model, self.device = model, torch.device("cpu")
        self.flat = torch.arange(8, dtype=torch.float32)
        self.grads = torch.full((8,), float(rank + 1))          # rank 0: ones, rank 1: twos
        self.params = []
        self.changed = 0
        self.serial, self.reduced_serial = 1, -1                 # not reduced stage by stage (no native step ran): Adam.step does it
    def grad_view(self, i): return self.grads[4 * i:4 * i + 4]
    def reduce_finish(self): pass
    def weights_changed(self): self.changed += 1
model = object()
tr = FakeTrainer(model)
for i in range(2):
    p = torch.nn.Parameter(tr.flat[4 * i:4 * i + 4].clone())
    p._hd_flat = (tr, 4 * i)
    p.grad = tr.grad_view(i)
    tr.params.append(p)
opt = O.Adam(tr.params, lr=2e-5)
opt.step()
assert torch.equal(tr.grads, torch.full((8,), 3.0)), tr.grads            # summed over the two ranks by ONE all-reduce of the flat buffer
assert calls == [(8, 2e-5, 1, 0.5)], calls                                # the mean is taken inside the Adam kernel: scale = 1 / world
assert tr.changed == 1
opt.zero_grad()
assert all(p.grad is None for p in tr.params)
opt.step()                                                               # nothing back-propagated since zero_grad: no launch, no collective
assert len(calls) == 1
dist.destroy_process_group()
print("ok")
"""


def test_training_gradient_all_reduce_world_size_2_gloo(tmp_path):
    """The N-rank glue of the training step (hicdiff_amd/optim.py) on CPU: one all-reduce of the flat gradient buffer per step, the 1/world
    folded into the Adam launch, no launch after zero_grad.  The HIP library is replaced by a recorder (the kernels are covered on the GPU)."""
    script = tmp_path / "train_worker.py"
    script.write_text(_TRAIN_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out.decode()
        assert b"ok" in out


_STAGED_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
from hicdiff_amd._training import StagedReducer

# a layout shaped like the hicedrn trainer's: [head | time MLP | per block (film, conv) | body_tail, tail]; conv slots of the blocks are final
# back to front (stage 0 with the tail, stage 1), everything else at the very end (stage 2)
sizes  = [7, 5, 11, 3, 13, 3, 11, 3, 13, 3, 17, 2]
stages = [2, 2, 2,  2, 1,  1, 2,  2, 0,  0, 0,  0]
slots, off = [], 0
for i, n in enumerate(sizes):
    slots.append((f"p{i}", off, (n,)))
    off += n
g = torch.Generator().manual_seed(100 + rank)
mine = torch.randn(off, generator=g)
both = sum(torch.randn(off, generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
grads = mine.clone()
red = StagedReducer(grads, slots, stages)
assert red.runs[0] == [(sum(sizes[:8]), 13 + 3 + 17 + 2)], red.runs[0]           # adjacent slots of one stage merge into one run
assert red.runs[1] == [(sum(sizes[:4]), 16)], red.runs[1]
assert [b.numel() for b in red.buckets] == [35, 16, sum(sizes) - 51]
order = []
red.launch(lambda k, stream: order.append(k))
assert order == [0, 1, 2] and red.pending                                          # stage order = the order the walk back finishes them
try:
    red.launch()
    raise SystemExit("a second launch before finish() must be refused")
except RuntimeError:
    pass
red.finish()
assert not red.pending
assert torch.equal(grads, both), (grads - both).abs().max()                       # every element summed exactly once
red.finish()                                                                       # idempotent
# a second step through the same buckets
grads.copy_(mine)
red.launch()
red.finish()
assert torch.equal(grads, both)
dist.destroy_process_group()
print("ok")
"""


def test_staged_gradient_reduction_world_size_2_gloo(tmp_path):
    """hicdiff_amd/_training.py StagedReducer on CPU: slots grouped by gradient stage into contiguous buckets, one asynchronous all-reduce per
    stage in stage order, sums scattered back so that every element of the flat buffer is summed exactly once; reusable step after step."""
    script = tmp_path / "staged_worker.py"
    script.write_text(_STAGED_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out.decode()
        assert b"ok" in out


def test_cli_flags_match_the_reference():
    """train.py / inference.py keep the reference's six flags, defaults and bool-typed -u quirk."""
    import inference
    import train
    a = inference.create_parser().parse_args([])
    assert (a.unspervised, a.batch_size, a.epoch, a.celline, a.celln, a.sigma) == (True, 64, 400, "Human", 1, 1)
    assert inference.create_parser().parse_args(["-u", "0"]).unspervised is True       # type=bool: any non-empty string
    assert inference.create_parser().parse_args(["-u", ""]).unspervised is False
    d = train.create_parser().parse_args([])
    assert (d.unspervised, d.batch_size, d.epoch, d.celline, d.celln, d.sigma) == (True, 64, 400, "Human", 1, 1)     # train.py:30-39
    assert train.create_parser().parse_args(["-u", ""]).unspervised is False
    t = train.create_parser().parse_args(["-b", "32", "-e", "3", "-l", "Dros", "-n", "2"])
    assert (t.batch_size, t.epoch, t.celline, t.celln) == (32, 3, "Dros", 2)
    assert train.create_parser().parse_args(["--arch", "unet", "--precision", "bf16"]).precision == "bf16"   # both networks train natively
    lq, hq = inference.synthetic_tiles(3, 16, 0.1, 7)
    assert lq.shape == hq.shape == (3, 1, 16, 16) and torch.equal(hq, hq.transpose(-1, -2)) and lq.abs().max() <= 1


_LAUNCH_WORKER = r"""
import json, os, sys, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1"
if len(sys.argv) > 1 and sys.argv[1] == "fail" and rank == 1:
    sys.exit(7)                                   # rank 0 would now wait in the collective forever
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t, op=dist.ReduceOp.MAX)
if rank == 0:
    print(json.dumps({"n_gpus": world, "max": t.item()}), flush=True)
dist.barrier()
dist.destroy_process_group()
"""


def test_launch_ranks_starts_n_processes_and_reaps_them_on_failure(tmp_path):
    """What `python bench.py --gpus N` does when it is not under torchrun (hicdiff_amd/sharding.py::launch_ranks):
    N fresh rank processes with the torchrun environment, rank 0's line on the caller's stdout, non-zero exit and no
    stragglers when a rank dies."""
    script = tmp_path / "w.py"
    script.write_text(_LAUNCH_WORKER)
    code = ("import sys; sys.path.insert(0, %r); from hicdiff_amd.sharding import launch_ranks; "
            "sys.exit(launch_ranks([sys.executable, %r] + sys.argv[1:], 2, timeout=100))" % (ROOT, str(script)))
    ok = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=150)
    assert ok.returncode == 0, ok.stderr.decode()
    assert json.loads(ok.stdout.decode().strip().splitlines()[-1]) == {"n_gpus": 2, "max": 2.0}
    t0 = time.time()
    bad = subprocess.run([sys.executable, "-c", code, "fail"], capture_output=True, timeout=150)
    assert bad.returncode == 7 and time.time() - t0 < 60       # the surviving rank was terminated, not waited for


def test_bench_self_launch_happens_before_any_gpu_use():
    """bench.py --gpus N without RANK must hand over to launch_ranks before importing the HIP library or calling torch.cuda."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    spawn = main.index("launch_ranks(")
    for needle in ("rank_env(args)", "L.load()", "torch.cuda", "bench_train(args)", "bench_tiles(args)"):
        assert main.index(needle) > spawn, needle


def test_fused_step_coefficients_follow_a_loaded_checkpoint():
    """A reference checkpoint carries the 13 schedule buffers and the reference samples from them (src/hicdiff.py:494-522):
    loading a 'linear' checkpoint into an object constructed with 'sigmoid' must switch the fused sampler's per-step scalars
    too, not only q_sample's buffers."""
    from hicdiff_amd.hicdiff import GaussianDiffusion, Unet
    net = Unet(16, dim_mults=(1, 2))
    lin = GaussianDiffusion(net, image_size=16, timesteps=1000, loss_type="l2", beta_schedule="linear")
    sig = GaussianDiffusion(net, image_size=16, timesteps=1000, loss_type="l2", beta_schedule="sigmoid")
    fields = ("sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1", "posterior_mean_coef2", "sigma")
    before = [getattr(sig._coef(500), f) for f in fields]
    want = [getattr(lin._coef(500), f) for f in fields]
    assert before != want
    sig.load_state_dict(lin.state_dict())
    assert [getattr(sig._coef(500), f) for f in fields] == want
    assert torch.equal(sig._host["alphas_cumprod"], lin.alphas_cumprod)          # DDIM reads this one
    # and the table is only re-read when a buffer changed
    assert sig._host is sig._host


def test_every_rank_runs_the_same_number_of_optimizer_steps():
    """train.py's batch plan: each training step ends in a collective, so ranks must take equally many batches for any tile
    count (the reference's DataLoader keeps the ragged tail; here it is filled up so the trainer's batch size never changes)."""
    import train
    for n, bs, world in ((256, 64, 8), (1000, 64, 8), (65, 64, 2), (7, 64, 3), (640, 64, 3), (64, 64, 1)):
        plans = [train.batch_plan(n, bs, r, world, True) for r in range(world)]
        assert len({len(p) for p in plans}) == 1 and len(plans[0]) >= 1, (n, bs, world)
        assert all(len(b) == bs for p in plans for b in p)
        seen = {i for p in plans for b in p for i in b}
        assert seen == set(range(n))                                           # every tile is trained on
        val = [train.batch_plan(n, bs, r, world, False) for r in range(world)]
        flat = sorted(i for p in val for b in p for i in b)
        assert flat == list(range(n))                                          # validation: each tile exactly once, ragged tail kept
    assert train.batch_plan(0, 64, 0, 2, True) == []


def test_precision_schedule_policy_on_the_host():
    """The samplers' precision schedule is host policy (hicdiff_amd/_diffusion.py:_coef -> hd_ddpm_coef.arith): long
    chains only, per network, switchable.  No GPU needed: the coefficients come from the host copies of the schedule buffers."""
    from hicdiff_amd import _lib as L
    from hicdiff_amd.hicdiff import GaussianDiffusion, Unet
    from hicdiff_amd.model.hicedrn_Diff import hicedrn_Diff
    u = GaussianDiffusion(Unet(16, dim_mults=(1, 2)), image_size=16, timesteps=1000, loss_type="l2", beta_schedule="linear")
    got = [u._coef(t).arith for t in (999, 750, 749, 500, 499, 0)]
    assert got == [L.HD_ARITH_F16W1, L.HD_ARITH_F16W1, L.HD_ARITH_F16W2, L.HD_ARITH_F16W2, L.HD_ARITH_F16W2_LOW, L.HD_ARITH_F16W2_LOW]
    u.late_band_low_f16 = False
    assert u._coef(100).arith == L.HD_ARITH_DEFAULT
    u.early_band_f16 = False
    assert {u._coef(t).arith for t in (999, 600, 100)} == {L.HD_ARITH_DEFAULT}
    short = GaussianDiffusion(Unet(16, dim_mults=(1, 2)), image_size=16, timesteps=50, loss_type="l2", beta_schedule="linear")
    assert {short._coef(t).arith for t in range(50)} == {L.HD_ARITH_DEFAULT}                      # 50-step chains keep three products
    h = GaussianDiffusion(hicedrn_Diff(number_resnet=1), image_size=16, timesteps=1000, loss_type="l2", beta_schedule="linear")
    assert [h._coef(t).arith for t in (999, 749, 0)] == [L.HD_ARITH_F16W1, L.HD_ARITH_F16W2, L.HD_ARITH_F16W2]   # hicedrn: the whole chain
    h.early_band_from = 0.5
    assert h._coef(0).arith == L.HD_ARITH_F16W2_LOW and h._coef(500).arith == L.HD_ARITH_F16W2
    # measured on the linear beta schedule only: the reference's default (sigmoid) and cosine keep three products at every step
    # (the linear bands cost them 1.0-2.4e-3: profiles/r04_s_*)
    for sched in ("sigmoid", "cosine"):
        g = GaussianDiffusion(Unet(16, dim_mults=(1, 2)), image_size=16, timesteps=1000, loss_type="l2", beta_schedule=sched)
        assert {g._coef(t).arith for t in range(0, 1000, 7)} == {L.HD_ARITH_DEFAULT}, sched
    for obj in ("pred_x0", "pred_v"):           # measured for the reference's objective only
        g = GaussianDiffusion(Unet(16, dim_mults=(1, 2)), image_size=16, timesteps=1000, loss_type="l2", beta_schedule="linear", objective=obj)
        assert {g._coef(t).arith for t in (999, 600, 100, 0)} == {L.HD_ARITH_DEFAULT}, obj
    sr3_like = GaussianDiffusion(Unet(16, dim_mults=(1, 2)), image_size=16, timesteps=2000, loss_type="l2", beta_schedule="linear")
    assert [sr3_like._coef(t).arith for t in (1999, 1500, 1499, 1000, 999)] == [L.HD_ARITH_F16W1, L.HD_ARITH_F16W1, L.HD_ARITH_F16W2, L.HD_ARITH_F16W2, L.HD_ARITH_F16W2_LOW]
    import ctypes as C
    assert C.sizeof(L.HdDdpmCoef) == 36 and C.sizeof(L.HdDdrmCoef) == 44                          # include/hicdiff_hip.h
