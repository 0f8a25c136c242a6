#!/bin/bash
# same-box A/B of the shipped library against contrastboundary_amd/lib/libcbl_amd_exp.so (another build: a kernel experiment, or the previous state):
# isolated kernel times (rocprofv3, in-order step) and the pipelined / one-at-a-time step.   usage: bash tools/gpu_exp_lib.sh "<kernel name regex>"
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
PAT=${1:-"kpconv_bwd|grouping_bwd|contrast_|knn_grid_wave|query_group"}
for lib in "" "$GRAFT_REPO_ROOT/contrastboundary_amd/lib/libcbl_amd_exp.so" "" "$GRAFT_REPO_ROOT/contrastboundary_amd/lib/libcbl_amd_exp.so"; do
  echo "== lib: ${lib:-shipped}"
  CBL_AMD_LIB=$lib bash tools/gpu_prof_any.sh exp 16 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --no-legs --no-gather-200k --no-overlap --no-pipeline --steps 100 --warmup 5 | grep -E "$PAT"
  CBL_AMD_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-legs --no-gather-200k --steps 200 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ms_per_step %.4f no_pipeline %.4f' % (d['ms_per_step'], d['no_pipeline']['ms_per_step']))"
done
