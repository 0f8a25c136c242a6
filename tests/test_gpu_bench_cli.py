"""bench.py as the driver runs it, on the GPU box: the JSON contract, the torchrun environment path, and the RCCL all-reduce leg (over a
one-rank group: the box has one GPU; the N > 1 rendezvous / timing logic is covered on gloo in test_distributed_gloo.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env=None):
    e = dict(os.environ)
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(key, None)
    e.update(env or {})
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines


def test_bench_json_contract_and_rccl_allreduce_leg():
    r, lines = _run([sys.executable, "bench.py", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--allreduce-single"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in out
    assert out["n_gpus"] == 1 and out["steps"] == 6 and out["warmup"] == 2 and out["scaling"] == "weak" and out["dtype"] == "f32"
    assert abs(out["value"] - 40960 * 6 / (out["ms_per_step"] * 6e-3)) < 1e-6 * out["value"]
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and 0.3 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert "backward" in out["config"]["workload"] and "forward_only" in out
    # the headline is the median of three timed regions, and the pipeline was held against one step at a time before it was timed (and overlaps)
    regions = out["timed_regions_ms_per_step"]
    assert len(regions) == 3 and abs(sorted(regions)[1] - out["ms_per_step"]) < 1e-3 * out["ms_per_step"]
    pc = out["pipeline_check"]
    # (the check's own figures, taken over 20 + 20 untimed steps; the six-step regions of THIS invocation are too short to hold the pipeline's gain to a bound)
    # a pipeline that still does not overlap after two rebuilds is reported (rebuilt == 2), not hidden: the line is the finding, this test only holds its shape
    assert pc["rebuilt"] in (0, 1, 2) and pc["pipelined_ms"] > 0 and pc["one_at_a_time_ms"] > 0, pc
    assert pc["pipelined_ms"] < 0.92 * pc["one_at_a_time_ms"] or pc["rebuilt"] == 2, pc
    sr = rf["search"]
    assert sr["pairs_visited"] > 36 * 40960 and 0 < sr["pairs_vs_brute_force"] < 0.05
    ar = out["grad_allreduce"]                                        # one flat fp32 buffer of the network's 7,800,497 gradients per step over RCCL
    assert ar["bytes"] == 4 * 7800497 and ar["ranks"] == 1 and ar["ms_per_step"] > 0 and ar["allreduce_alone_ms"] > 0
    # the default line carries BASELINE's other single-GPU configurations as legs (each measured in a process of its own)
    pt, cn = out["pt_block"], out["convnet"]
    assert "error" not in pt and "error" not in cn, (pt, cn)
    assert "PointTransformer" in pt["metric"] and pt["ms_per_step"] > 0 and abs(pt["value"] - 40960 / (pt["ms_per_step"] * 1e-3)) < 1e-6 * pt["value"]
    assert pt["roofline"]["frac_of_f32_mfma_peak"] > 0 and pt["roofline"]["frac"] > 0
    assert "ConvNet" in cn["metric"] and cn["ms_per_step"] > 0 and cn["roofline"]["frac"] > 0 and "adaptive_weight" in cn["roofline"]


def test_bench_under_torchrun_environment():
    r, lines = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29577",
                     "bench.py", "--gpus", "1", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-extra"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 1


def test_bench_refuses_two_gpus_on_a_one_gpu_box():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("more than one device here")
    r, lines = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2 and not lines and "device(s) visible" in r.stderr
