#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {
    for rep in 1 2 3 4; do
        CBL_PIPELINE_LAYOUT=$1 CBL_PIPELINE_SLOTS=$2 CBL_PIPELINE_TUNE=0 timeout 300 python bench.py --block pt --no-cpu-baseline --no-extra --no-legs --no-gather-200k --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'], end=' ')"
    done; echo " <- pt $1 slots $2"
}
run split_fwd 2
run split_fwd_t36_first 2
run split_fwd_t36_first 3
run split_t36_first 3
run split_side_late 3
