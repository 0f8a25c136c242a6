#!/bin/bash
# parity tests of the headline's kernels, isolated kernel times (rocprofv3, in-order step), and the pipelined step under both layouts
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/hl
timeout 900 python -m pytest tests/test_gpu_bench_step.py tests/test_gpu_transpose.py tests/test_gpu_cbl.py tests/test_gpu_local_aggregation.py -x -q 2>&1 | tail -5
bash tools/gpu_prof_any.sh hl 12 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --no-legs --no-gather-200k --no-overlap --no-pipeline --steps 100 --warmup 5
for lay in "tables 2" "split 3" "tables 2" "split 3"; do
    set -- $lay
    CBL_PIPELINE_LAYOUT=$1 CBL_PIPELINE_SLOTS=$2 timeout 300 python bench.py --no-cpu-baseline --no-extra --no-legs --no-gather-200k --steps 200 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 ms_per_step %.4f' % d['ms_per_step'])"
done
timeout 300 python bench.py --no-cpu-baseline --no-extra --no-legs --no-gather-200k --no-pipeline --steps 200 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_pipeline ms_per_step %.4f' % d['ms_per_step'])"
