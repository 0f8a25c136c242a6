#!/bin/bash
# spread of the pipelined step over processes (one box): "<block> <tune> <runs>"
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {
    for rep in $(seq 1 $3); do
        CBL_PIPELINE_TUNE=$2 timeout 300 python bench.py --block $1 --no-cpu-baseline --no-extra --no-legs --no-gather-200k --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'], end=' ')"
    done; echo " <- $1 tune $2"
}
run kpconv 0 14
run kpconv 1 14
run pt 0 10
run pt 1 10
