#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/cv
timeout 900 python -m pytest tests/test_gpu_tfops.py tests/test_gpu_bench_convnet.py -x -q 2>&1 | tail -8
timeout 600 python bench.py --workload convnet --no-cpu-baseline --no-extra > gpurun_out/cv/bench.json 2> gpurun_out/cv/bench.err; echo "rc=$?"; tail -3 gpurun_out/cv/bench.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/cv/bench.json").read().strip().splitlines()[-1])
print("value ms", d["ms_per_step"], "in-order", d["no_pipeline"], "pipelined", d["pipelined"])
P
