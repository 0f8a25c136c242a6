#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {
    for rep in 1 2 3 4; do
        CBL_PIPELINE_LAYOUT=$1 CBL_PIPELINE_SLOTS=$2 CBL_PIPELINE_TUNE=0 timeout 300 python bench.py --block ${3:-pt} --no-cpu-baseline --no-extra --no-legs --no-gather-200k --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'], end=' ')"
    done; echo " <- ${3:-pt} $1 slots $2"
}
run alt_bwd 4
run alt_main 4
run alt_main 3
run alt_main 4 kpconv
