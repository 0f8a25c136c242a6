"""CPU, world_size 2, gloo: the N>1 logic of the hot path — disjoint scene sharding, no data-path collective, whole-job
throughput = sum(units) / max(time).  (The kernels themselves need a GPU; ranks here run the host logic only.)"""
import os
import socket

import torch
import torch.multiprocessing as mp


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


def test_bench_refuses_more_gpus_than_devices():
    r = _bench("--gpus", "2", "--steps", "1", "--warmup", "0")        # no GPU here: must fail loudly, never print n_gpus
    assert r.returncode == 2 and "device(s) visible" in r.stderr and "{" not in r.stdout


def test_bench_refuses_a_world_that_is_not_gpus():
    r = _bench("--gpus", "4", "--host-dry-run", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    assert r.returncode == 2 and "WORLD_SIZE=2 but --gpus 4" in r.stderr
