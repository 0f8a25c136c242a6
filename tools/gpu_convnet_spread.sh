#!/bin/bash
# parity of the pyramid builders / backward forms and the spread of the ConvNet step over processes (one box)
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_tfops.py tests/test_gpu_bench_convnet.py -x -q 2>&1 | tail -2
for rep in 1 2 3 4 5 6; do
    timeout 300 python bench.py --workload convnet --no-cpu-baseline --no-extra --steps 30 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f/%.3f' % (d['pipelined']['ms_per_step'], d['no_pipeline']['ms_per_step']), end=' ')"
done; echo " <- pipelined / in order"
