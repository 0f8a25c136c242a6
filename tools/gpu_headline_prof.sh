#!/bin/bash
# isolated kernel times of the headline step (rocprofv3, in-order step)
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_prof_any.sh hl 24 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --no-legs --no-gather-200k --no-overlap --no-pipeline --steps 100 --warmup 5
