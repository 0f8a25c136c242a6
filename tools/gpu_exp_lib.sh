#!/bin/bash
# isolated kernel times with the shipped library and with contrastboundary_amd/lib/libcbl_amd_exp.so (a kernel experiment build), same box
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
for lib in "" "$GRAFT_REPO_ROOT/contrastboundary_amd/lib/libcbl_amd_exp.so" "" "$GRAFT_REPO_ROOT/contrastboundary_amd/lib/libcbl_amd_exp.so"; do
  echo "== lib: ${lib:-shipped}"
  CBL_AMD_LIB=$lib bash tools/gpu_prof_any.sh exp 14 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --no-legs --no-gather-200k --no-overlap --no-pipeline --steps 100 --warmup 5 | grep "kpconv_bwd\|grouping_bwd\|contrast_gather\|knn_grid_wave"
done
