#!/bin/bash
# A/B of hotpath.Pipeline layouts in ONE gpurun call (box-to-box spread is larger than the differences): "<layout> <slots> <tune>" x repetitions
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {
    for rep in 1 2 3 4 5 6; do
        CBL_PIPELINE_LAYOUT=$1 CBL_PIPELINE_SLOTS=$2 CBL_PIPELINE_TUNE=$3 timeout 300 python bench.py --no-cpu-baseline --no-extra --no-legs --no-gather-200k --steps 200 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'], end=' ')"
    done; echo " <- $1 slots $2 tune $3"
}
run split 3 0
run split_side_late 3 0
run split_t36_first 3 0
run split_side_late 2 0
run split_t36_first 4 0
