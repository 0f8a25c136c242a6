"""CPU, world_size 2, gloo: the N>1 logic of the hot path — disjoint scene sharding, no data-path collective, whole-job
throughput = sum(units) / max(time).  (The kernels themselves need a GPU; ranks here run the host logic only.)"""
import os
import socket

import pytest

import torch
import torch.multiprocessing as mp


def rendezvous_retry(fn):
    """the tests below probe a free port and rendezvous on it a moment later (two processes, or bench.py's own launcher): if anything else on the host takes
    the port in between, the run fails for a reason that is not the code's — such a test is run once more before it counts as failed"""
    import functools

    @functools.wraps(fn)
    def wrapper(*a, **k):
        try:
            return fn(*a, **k)
        except Exception:                                                  # noqa: BLE001 - second attempt decides
            return fn(*a, **k)
    return wrapper


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from contrastboundary_amd import distributed as D
    w, r, _ = D.init("gloo")
    scenes = D.shard_scenes(11, r, w)
    # every rank "processes" its scenes: 40960 points each, rank 1 is slower
    units, elapsed = 40960.0 * len(scenes), 0.5 + 0.25 * r
    D.barrier()
    thr, total, worst = D.aggregate_throughput(units, elapsed)
    out.put((r, scenes, thr, total, worst))
    import torch.distributed as dist
    dist.destroy_process_group()


@rendezvous_retry
def test_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (r0, s0, thr0, tot0, w0), (r1, s1, thr1, tot1, w1) = res
    assert sorted(s0 + s1) == list(range(11)) and not set(s0) & set(s1)          # disjoint cover
    assert tot0 == tot1 == 40960.0 * 11 and w0 == w1 == 0.75                      # sum of units, max of times
    assert abs(thr0 - 40960.0 * 11 / 0.75) < 1e-6 and thr0 == thr1


def test_single_process_is_identity():
    from contrastboundary_amd import distributed as D
    assert D.shard_scenes(5, 0, 1) == [0, 1, 2, 3, 4]
    assert D.aggregate_throughput(100.0, 2.0) == (50.0, 100.0, 2.0)


# ---- bench.py's own launcher path (`--gpus N` outside a torchrun environment), world 2 over gloo on the CPU -------------------------
def _bench(*argv, env=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(key, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), capture_output=True, text=True, timeout=300, env=e, cwd=root)


@rendezvous_retry
def test_bench_spawns_its_ranks_and_reports_the_joined_world():
    import json
    r = _bench("--gpus", "2", "--host-dry-run", "--steps", "5", "--warmup", "1", "--allreduce-floats", "4096")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 1 and out["scaling"] == "weak"
    # 5 steps of >= 2 ms each between the barriers; value = units of BOTH ranks / max time
    assert out["ms_per_step"] >= 2.0 and abs(out["value"] - 40960 * 5 * 2 / (out["ms_per_step"] * 5e-3)) < 1e-6 * out["value"]
    # the gradient all-reduce leg ran over the process group: ones summed over 2 ranks and averaged stay 1
    assert out["grad_allreduce"]["bytes"] == 4 * 4096 and out["grad_allreduce"]["checksum"] == 1.0
    # who took part, and how evenly: the group's size and backend, the fastest / slowest rank's own time per step
    rk = out["ranks"]
    assert rk["rccl_ranks"] == 2 and rk["backend"] == "gloo" and 2.0 <= rk["ms_per_step_min_rank"] <= rk["ms_per_step_max_rank"] <= out["ms_per_step"] + 1e-6


@rendezvous_retry
@pytest.mark.parametrize("extra,what", [(["--workload", "convnet"], "ConvNet"), (["--block", "pt"], "pt block")])
def test_bench_launcher_with_the_other_workloads(extra, what):
    """`bench.py --gpus N --workload convnet` / `--block pt`: the same spawn / rendezvous / timing path as the headline (the driver's scaling runs
    use the default line; these are one command away from a curve of their own)"""
    import json
    r = _bench("--gpus", "2", "--host-dry-run", "--steps", "3", "--warmup", "1", "--no-allreduce", *extra)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"]["rccl_ranks"] == 2 and what in out["config"]["workload"]
    if extra[0] == "--workload":
        assert abs(out["value"] - 200000 * 3 * 2 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]          # the ConvNet scene's default size


def _cold_build_worker(i, libdir, q):
    """one of N ranks that all find the library stale at start-up (torchrun with cold ranks): every one calls build(); the file lock serialises them"""
    import time
    from contrastboundary_amd import build as B
    t = time.time()
    so = B.build()
    q.put((i, os.path.exists(so), time.time() - t))


def test_eight_cold_ranks_build_the_library_once():
    """build.py's flock: 8 processes that all call build() at once on a stale library (a header touched) — one compiles / links, the others wait and
    find it fresh; nobody sees a half-written .so (objects and library are renamed into place)"""
    import ctypes
    from contrastboundary_amd import build as B
    B.build()
    hdr = os.path.join(B.CSRC, "wave_ops.h")
    st = os.stat(hdr)
    try:
        os.utime(hdr)                                                 # every object that includes a header is stale now ... as far as mtimes go
        assert B.is_stale()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_cold_build_worker, args=(i, B.LIBDIR, q)) for i in range(8)]
        [p.start() for p in procs]
        res = [q.get(timeout=900) for _ in range(8)]
        [p.join(60) for p in procs]
        assert all(p.exitcode == 0 for p in procs) and all(ok for _, ok, _ in res)
        assert not B.is_stale()
        L = ctypes.CDLL(B.SO)                                         # loads: a complete library
        L.cbl_version.restype = ctypes.c_char_p
        assert b"cbl_amd" in L.cbl_version()
    finally:
        os.utime(hdr, (st.st_atime, st.st_mtime))
        B.build()


def test_bench_refuses_more_gpus_than_devices():
    r = _bench("--gpus", "2", "--steps", "1", "--warmup", "0")        # no GPU here: must fail loudly, never print n_gpus
    assert r.returncode == 2 and "device(s) visible" in r.stderr and "{" not in r.stdout


def test_bench_refuses_a_world_that_is_not_gpus():
    r = _bench("--gpus", "4", "--host-dry-run", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    assert r.returncode == 2 and "WORLD_SIZE=2 but --gpus 4" in r.stderr


# ---- data-parallel training: the gradient reducer (DDP's role, train.py:181-185) and tools/bench_model.py's rank logic ---------------
class _Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.net = torch.nn.Sequential(torch.nn.Linear(5, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
        self.unused = torch.nn.Linear(4, 4)                           # never takes part in the forward: its bucket must still be reduced (zeros)

    def forward(self, x):
        return self.net(x)


def _tiny_model(seed):
    torch.manual_seed(seed)
    return _Tiny()


def _rank_batch(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(32, 5, generator=g), torch.randn(32, 3, generator=g)


def _reducer_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from contrastboundary_amd import distributed as D
    D.init("gloo")
    model = _tiny_model(7 + rank)                                     # different per rank: broadcast_parameters equalises
    D.broadcast_parameters(model)
    red = D.GradientReducer(model.parameters(), bucket_bytes=128)     # 32 floats per bucket -> several buckets, some spanning parameters
    launched = []
    orig = red._launch
    red._launch = lambda k: (launched.append(k), orig(k))[1]
    x, y = _rank_batch(rank)
    grads = []
    for it in range(2):                                               # twice: the views must survive zero_grad, the bucket order must repeat
        red.zero_grad()
        ((model(x) - y) ** 2).mean().backward()
        red.finish()
        grads.append(torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone())
    views = all(p.grad.data_ptr() >= red.flat.data_ptr() and p.grad.data_ptr() < red.flat.data_ptr() + 4 * red.flat.numel() for p in model.parameters())
    out.put((rank, grads[0], grads[1], launched, len(red.buckets), views))
    import torch.distributed as dist
    dist.destroy_process_group()


@rendezvous_retry
def test_gradient_reducer_averages_real_gradients_over_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    # single-process truth: rank 0's weights, each rank's batch, gradients averaged
    ref = []
    for rank in range(2):
        m = _tiny_model(7)
        x, y = _rank_batch(rank)
        ((m(x) - y) ** 2).mean().backward()
        ref.append(torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in m.parameters()]))
    want = (ref[0] + ref[1]) / 2
    for rank, g0, g1, launched, nbuckets, views in res:
        assert torch.allclose(g0, want, rtol=1e-6, atol=1e-7) and torch.equal(g0, g1)
        assert launched == list(range(nbuckets)) * 2 and nbuckets >= 3         # strictly in bucket order, every step, on every rank
        assert views                                                          # .grad stayed a view of the flat buffer
    assert torch.equal(res[0][1], res[1][1])                                  # bitwise the same on both ranks


def test_gradient_reducer_single_process_is_identity():
    from contrastboundary_amd import distributed as D
    m = _tiny_model(3)
    red = D.GradientReducer(m.parameters(), bucket_bytes=128)
    x, y = _rank_batch(0)
    red.zero_grad(); ((m(x) - y) ** 2).mean().backward(); red.finish()
    got = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    m2 = _tiny_model(3)
    ((m2(x) - y) ** 2).mean().backward()
    want = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in m2.parameters()])
    assert torch.equal(got, want)


@rendezvous_retry
def test_bench_model_deals_scenes_and_keeps_replicas_identical():
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(key, None)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_model.py"), "--gpus", "2", "--host-dry-run", "--scenes", "3", "--steps", "4", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=e, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scenes_of_rank0"] == [0, 2, 4]          # 6 scenes dealt round-robin over 2 ranks
    assert out["replicas_identical"] is True                                   # started different, broadcast + averaged gradients keep them equal
    assert out["grad_allreduce"]["ranks"] == 2 and out["grad_allreduce"]["buckets"] >= 2
    assert out["ranks"]["rccl_ranks"] == 2 and out["ranks"]["ms_per_step_min_rank"] <= out["ranks"]["ms_per_step_max_rank"]


# ---- the data-parallel trainer: criterion parameters averaged with the model's, buffers re-broadcast per step (DDP's roles, train.py:181-189) ----
class _Crit(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.scale = torch.nn.Parameter(torch.ones(3))                # a trainable criterion parameter (the CBL head's `project` MLP in the reference)
        self.bn = torch.nn.BatchNorm1d(3)                             # ... and buffers of its own

    def forward(self, out, y):
        return ((self.bn(out) * self.scale - y) ** 2).mean().unsqueeze(0)


class _BnNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.net = torch.nn.Sequential(torch.nn.Linear(5, 8), torch.nn.BatchNorm1d(8), torch.nn.ReLU(), torch.nn.Linear(8, 3))

    def forward(self, x):
        return self.net(x)


def _trainer_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from contrastboundary_amd import distributed as D, train_step
    D.init("gloo")
    torch.manual_seed(50 + rank)                                      # different initial weights per rank
    model, crit = _BnNet(), _Crit()
    opt = torch.optim.SGD(list(model.parameters()) + list(crit.parameters()), lr=0.1)
    seen = []
    tr = train_step.DataParallelTrainer(model, crit, opt, bucket_bytes=64, forward_loss=lambda m, c, x, y: c(m(x), y))
    x, y = _rank_batch(rank)
    for _ in range(3):
        tr.step(x, y)
        seen.append(torch.cat([b.reshape(-1).float() for m in (model, crit) for b in m.buffers()]).clone())
    D.broadcast_buffers([model, crit])                                # what the next step would start from
    params = torch.cat([p.detach().reshape(-1) for m in (model, crit) for p in m.parameters()])
    bufs = torch.cat([b.reshape(-1).float() for m in (model, crit) for b in m.buffers()])
    tr.reducer.remove(); n0 = len(tr.reducer.handles); tr.reducer.rehook(); n1 = len(tr.reducer.handles)
    out.put((rank, params.numpy().copy(), bufs.numpy().copy(), len(tr.reducer.params), seen[-1].numpy().copy(), n0, n1))
    import torch.distributed as dist
    dist.destroy_process_group()


@rendezvous_retry
def test_trainer_averages_criterion_parameters_and_broadcasts_buffers():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (_, p0, b0, n0, own0, h0, h1), (_, p1, b1, n1, own1, _, _) = res
    assert n0 == n1 == 9                                              # 6 model parameters and the criterion's 3 (scale, bn.weight, bn.bias) in one reducer
    import numpy as np
    assert np.array_equal(p0, p1)                                     # every parameter — the criterion's included — identical on both ranks after 3 steps
    assert np.array_equal(b0, b1)                                     # after the broadcast: rank 0's running statistics everywhere
    assert not np.array_equal(own0, own1)                             # ... which the ranks' own batches had moved apart within the step
    assert h0 == 0 and h1 == n0                                       # hooks can be re-registered after a capture removed them


# ---- round 5: the flat state (distributed.FlatState + PackedGradientReducer): parameters / gradients / buffers as views of flat tensors, gradients
#      packed behind the backward, ONE optimizer kernel, ONE buffer broadcast — and the collective issued on a ONE-rank group too --------------------
def _flat_trainer_worker(rank, world, port, out, flat):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from contrastboundary_amd import distributed as D, train_step
    D.init("gloo")
    if world == 1:
        import torch.distributed as dist
        dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    torch.manual_seed(50 + rank)
    model, crit = _BnNet(), _Crit()
    opt = torch.optim.SGD(list(model.parameters()) + list(crit.parameters()), lr=0.1, momentum=0.9, weight_decay=1e-3)
    tr = train_step.DataParallelTrainer(model, crit, opt, bucket_bytes=64, forward_loss=lambda m, c, x, y: c(m(x), y), flat=flat)
    issued = []
    if flat:
        import torch.distributed as dist
        orig = dist.all_reduce
        dist.all_reduce = lambda *a, **k: (issued.append(1), orig(*a, **k))[1]
    x, y = _rank_batch(rank)
    for it in range(4):
        if it == 2:
            for g in opt.param_groups:                                # an LR scheduler steps the CALLER's optimizer: the flat twin must follow
                g["lr"] = 0.05
        tr.step(x, y)
    params = torch.cat([p.detach().reshape(-1) for m in (model, crit) for p in m.parameters()])
    bufs = torch.cat([b.reshape(-1).float() for m in (model, crit) for b in m.buffers()])
    aligned = views = True
    if flat:
        st = tr.state
        lo, hi = st.flat_param.data_ptr(), st.flat_param.data_ptr() + 4 * st.flat_param.numel()
        views = all(lo <= p.data_ptr() < hi for m in (model, crit) for p in m.parameters())
        aligned = all((p.data_ptr() - lo) % 256 == 0 for m in (model, crit) for p in m.parameters())
        fb = st.flat_buffers["float"]
        views = views and all(fb.data_ptr() <= b.data_ptr() < fb.data_ptr() + 4 * fb.numel() for m in (model, crit) for b in m.buffers() if b.is_floating_point())
    out.put((rank, params.numpy().copy(), bufs.numpy().copy(), views, aligned, len(issued), len(tr.reducer.buckets)))
    import torch.distributed as dist
    dist.destroy_process_group()


def _run_flat(world, flat):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flat_trainer_worker, args=(r, world, port, q, flat)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    return res


@rendezvous_retry
def test_flat_state_trainer_equals_the_hook_trainer_over_two_ranks():
    """four steps (momentum, weight decay, a learning-rate change between steps): the packed-gradient / one-tensor-optimizer step lands on the SAME parameters
    as round 4's per-tensor step, replicas identical, parameters and buffers views of the flat tensors on 256-byte boundaries"""
    import numpy as np
    a, b = _run_flat(2, True), _run_flat(2, False)
    assert np.array_equal(a[0][1], a[1][1]) and np.array_equal(b[0][1], b[1][1])          # replicas identical, either layout
    np.testing.assert_allclose(a[0][1], b[0][1], rtol=1e-6, atol=1e-7)                   # same trajectory (elementwise update: flat == per tensor)
    np.testing.assert_allclose(a[0][2], b[0][2], rtol=1e-6, atol=1e-7)
    assert all(r[3] and r[4] for r in a)
    assert all(r[5] == 4 * r[6] and r[6] >= 2 for r in a)                                # every bucket all-reduced in every step


@rendezvous_retry
def test_flat_state_issues_the_collective_on_a_one_rank_group():
    """a one-rank run must exercise packing + collective + optimizer together — the all-reduce is issued whenever a process group exists"""
    (rank, params, bufs, views, aligned, issued, nbuckets), = _run_flat(1, True)
    assert views and aligned and issued == 4 * nbuckets and nbuckets >= 2
    # ... and the hook reducer does the same on a one-rank group
    import torch.distributed as dist
    from contrastboundary_amd import distributed as D
    port = _free_port()
    dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        m = _tiny_model(3)
        red = D.GradientReducer(m.parameters(), bucket_bytes=128)
        calls = []
        orig = dist.all_reduce
        dist.all_reduce = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            x, y = _rank_batch(0)
            red.zero_grad(); ((m(x) - y) ** 2).mean().backward(); red.finish()
        finally:
            dist.all_reduce = orig
        assert red.grouped and red.world == 1 and len(calls) == len(red.buckets)
    finally:
        dist.destroy_process_group()


def test_flat_state_single_process_matches_the_plain_optimizer():
    """no process group: FlatState + its one-tensor optimizer follow torch.optim.SGD over the separate tensors step for step, momentum carried over from
    steps the caller's optimizer took before the state was built (the graph step's warm-up)"""
    from contrastboundary_amd import distributed as D
    torch.manual_seed(5)
    a, b = _BnNet(), _BnNet()
    b.load_state_dict(a.state_dict())
    oa = torch.optim.SGD(a.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-3)
    ob = torch.optim.SGD(b.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-3)
    x, y = _rank_batch(0)
    for m, o in ((a, oa), (b, ob)):                                   # one plain step on both: `ob` holds momentum when the state takes over
        o.zero_grad(); ((m(x) - y) ** 2).mean().backward(); o.step()
    st = D.FlatState([b], ob)
    st.flat_optimizer(ob)
    red = D.PackedGradientReducer(st, bucket_bytes=64)
    assert not red.grouped
    for _ in range(3):
        oa.zero_grad(); ((a(x) - y) ** 2).mean().backward(); oa.step()
        red.zero_grad(); ((b(x) - y) ** 2).mean().backward(); red.finish(); st.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-6, atol=1e-7)
    for ba, bb in zip(a.buffers(), b.buffers()):
        assert torch.allclose(ba.float(), bb.float(), rtol=1e-6, atol=1e-7)
    assert b.state_dict().keys() == a.state_dict().keys()


def test_flat_state_checkpoint_round_trip_and_adam_state():
    """the reference checkpoints 'optimizer': optimizer.state_dict() and resumes with load_state_dict (/root/reference/pytorch/tool/train.py:216,292): with the
    flat state the live momentum sits in the one-tensor optimizer — FlatState.state_dict() writes it through in the caller's per-parameter format, and
    load_state_dict() brings a checkpoint back into the flat tensors.  Adam's exp_avg / exp_avg_sq / step are carried like SGD's momentum_buffer."""
    from contrastboundary_amd import distributed as D
    x, y = _rank_batch(0)
    for make in (lambda ps: torch.optim.SGD(ps, lr=0.1, momentum=0.9, weight_decay=1e-3), lambda ps: torch.optim.Adam(ps, lr=1e-2)):
        torch.manual_seed(7)
        a, b = _BnNet(), _BnNet()
        b.load_state_dict(a.state_dict())
        oa, ob = make(a.parameters()), make(b.parameters())
        for m, o in ((a, oa), (b, ob)):                               # two plain steps: both optimizers hold state when the flat one takes over
            for _ in range(2):
                o.zero_grad(); ((m(x) - y) ** 2).mean().backward(); o.step()
        st = D.FlatState([b], ob)
        st.flat_optimizer(ob)
        red = D.PackedGradientReducer(st, bucket_bytes=64)
        for _ in range(2):
            oa.zero_grad(); ((a(x) - y) ** 2).mean().backward(); oa.step()
            red.zero_grad(); ((b(x) - y) ** 2).mean().backward(); red.finish(); st.step()
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-7)
        # checkpoint: the per-parameter state of the flat run equals the plain run's
        sd, ref = st.state_dict(), oa.state_dict()
        assert sd["state"].keys() == ref["state"].keys()
        for k in ref["state"]:
            for name, v in ref["state"][k].items():
                got = sd["state"][k][name]
                assert torch.allclose(torch.as_tensor(got).float(), torch.as_tensor(v).float(), rtol=1e-5, atol=1e-7), (type(oa).__name__, k, name)
        # resume: a fresh model + optimizer + flat state loaded from the checkpoint continues like the plain run
        import copy
        c = _BnNet(); c.load_state_dict(copy.deepcopy(b.state_dict()))
        oc = make(c.parameters())
        sc = D.FlatState([c], oc); sc.flat_optimizer(oc)
        sc.load_state_dict(copy.deepcopy(sd))
        rc = D.PackedGradientReducer(sc, bucket_bytes=64)
        oa.zero_grad(); ((a(x) - y) ** 2).mean().backward(); oa.step()
        rc.zero_grad(); ((c(x) - y) ** 2).mean().backward(); rc.finish(); sc.step()
        for pa, pc in zip(a.parameters(), c.parameters()):
            assert torch.allclose(pa, pc, rtol=1e-5, atol=1e-7)


def test_flat_state_leaves_parameters_without_a_gradient_alone():
    """torch's optimizers skip a parameter whose .grad is None (the reference's DDP runs with find_unused_parameters, train.py:184): an unused head must not
    decay or coast on its momentum under the fused flat step either"""
    from contrastboundary_amd import distributed as D
    torch.manual_seed(9)

    class TwoHeads(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.body = torch.nn.Linear(6, 16); self.used = torch.nn.Linear(16, 4); self.unused = torch.nn.Linear(16, 4)

        def forward(self, x):
            return self.used(torch.relu(self.body(x)))
    a, b = TwoHeads(), TwoHeads()
    b.load_state_dict(a.state_dict())
    oa = torch.optim.SGD(a.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-2)
    ob = torch.optim.SGD(b.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-2)
    st = D.FlatState([b], ob); st.flat_optimizer(ob)
    red = D.PackedGradientReducer(st, bucket_bytes=64)
    x = torch.randn(32, 6); y = torch.randn(32, 4)
    before = b.unused.weight.detach().clone()
    for _ in range(3):
        oa.zero_grad(); ((a(x) - y) ** 2).mean().backward(); oa.step()
        red.zero_grad(); ((b(x) - y) ** 2).mean().backward(); red.finish(); st.step()
    assert torch.equal(b.unused.weight, before) and torch.equal(a.unused.weight, before)
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-6, atol=1e-7)
