#!/bin/bash
# A/B of hotpath.Pipeline layouts in ONE gpurun call (box-to-box spread is larger than the differences): "<block> <layout> <slots>", six processes each.
# The numbers of DESIGN.md 6.4 come from runs of this script.
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {
    for rep in 1 2 3 4 5 6; do
        CBL_PIPELINE_LAYOUT=$2 CBL_PIPELINE_SLOTS=$3 timeout 300 python bench.py --block $1 --no-cpu-baseline --no-extra --no-legs --no-gather-200k --steps 200 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'], end=' ')"
    done; echo " <- $1 $2 slots $3"
}
run kpconv tables 2
run kpconv split 3
run kpconv split_t36_first 3
run kpconv split_side_late 3
run kpconv three 3
run pt tables 2
run pt split_fwd_t36_first 3
run pt alt_bwd 4
