#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bench_step.py -x -q 2>&1 | tail -15
for g in 0 1; do
timeout 600 python bench.py --group $g --no-cpu-baseline 2>gpurun_out/err_$g.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['forward_only']['ms_per_step'], d['config']['issue'][:90])
print(d['roofline']['stage_ms']); print(d['roofline']['mfma_kpconv']['launch_us'], d['roofline']['launch_us'], d['roofline']['scatter_k4']['launch_us'])"; tail -3 gpurun_out/err_$g.txt
done
