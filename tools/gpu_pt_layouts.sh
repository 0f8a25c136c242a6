#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_bench_step_pt.py tests/test_gpu_bench_step.py -x -q 2>&1 | tail -3
for lay in "tables 2" "split 2" "split_fwd 2" "split_fwd 3"; do
    set -- $lay
    CBL_PIPELINE_LAYOUT=$1 CBL_PIPELINE_SLOTS=$2 timeout 300 python bench.py --block pt --no-cpu-baseline --no-legs --no-gather-200k --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pt $1 $2 ms_per_step %.4f' % d['ms_per_step'], 'pipelined', d.get('pipelined',{}).get('ms_per_step'), 'no_pipeline', d.get('no_pipeline',{}).get('ms_per_step'))"
done
for lay in "tables 2" "split 3" "split_fwd 2" "split_fwd 3" "split_fwd 3"; do
    set -- $lay
    CBL_PIPELINE_LAYOUT=$1 CBL_PIPELINE_SLOTS=$2 timeout 300 python bench.py --no-cpu-baseline --no-extra --no-legs --no-gather-200k --steps 200 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kpconv $1 $2 ms_per_step %.4f' % d['ms_per_step'])"
done
