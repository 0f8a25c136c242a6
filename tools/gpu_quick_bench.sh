#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/quick
timeout 600 python -m pytest tests/test_gpu_bench_step_pt.py tests/test_gpu_bench_step.py tests/test_gpu_bench_cli.py tests/test_gpu_hotpath.py -x -q 2>&1 | tail -3
timeout 600 python bench.py > gpurun_out/quick/bench.json 2> gpurun_out/quick/bench.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open("gpurun_out/quick/bench.json").read().strip().splitlines()[-1])
print("headline %.4f no_pipeline %.4f fwd_only %.4f" % (d["ms_per_step"], d["no_pipeline"]["ms_per_step"], d["forward_only"]["ms_per_step"]))
print("pt_block", d["pt_block"].get("ms_per_step"), d["pt_block"].get("issue","")[:200])
print("convnet", d["convnet"].get("ms_per_step"))
print("stage_ms", d["roofline"]["stage_ms"])
print("gather frac", d["roofline"]["frac"], "200k", d["roofline"]["gather_200k"]["frac"], "k4", d["roofline"]["scatter_k4"]["frac"], d["roofline"]["scatter_k4"]["launch_us"])
print(d["config"]["issue"][:600])
P
