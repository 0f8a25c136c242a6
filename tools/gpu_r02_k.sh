#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bench_step.py -x -q 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-extra 2>gpurun_out/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'])"; tail -2 gpurun_out/err.txt
timeout 600 python bench.py --no-cpu-baseline --no-extra --forward-only 2>gpurun_out/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fwd', d['ms_per_step'], d['value'])"
bash tools/exp/prof_one.sh g $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 10 > /dev/null
