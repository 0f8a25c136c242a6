#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_local_aggregation.py tests/test_gpu_bench_step.py -x -q 2>&1 | tail -3
timeout 600 python bench.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['forward_only']['ms_per_step'])
print(d['roofline']['stage_ms']); print(d['roofline']['mfma_kpconv']); print(d['roofline']['launch_us'], d['roofline']['scatter_k4']['launch_us'])"
