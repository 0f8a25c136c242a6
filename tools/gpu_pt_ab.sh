#!/bin/bash
# same-box A/B of the fused attention layer's kernels across builds of the library: bash tools/gpu_pt_ab.sh <lib.so | ""> ...   ("" = the in-tree build)
# per build: rocprofv3 kernel averages of tools/pt_layer_time.py (40960, 16, 64) and (40960, 8, 32), then the pipelined --block pt step
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  tag=$(basename "${lib:-intree}" .so)
  echo "=== build: ${lib:-in-tree}  ${CBL_PT_NARROW_ROWS:+narrow rows $CBL_PT_NARROW_ROWS}"
  for shape in "40960 16 64" "40960 8 32"; do
    CBL_AMD_LIB=$lib bash tools/gpu_prof_any.sh ab_$tag 40 python $GRAFT_REPO_ROOT/tools/pt_layer_time.py $shape | grep -E "^pt_|^triple" | awk -v s="$shape" '{printf "  [%s] %-44s %8s us\n", s, $1" "$2" "$3, $(NF-5)}' | sed 's/calls//'
  done
  CBL_AMD_LIB=$lib timeout 300 python bench.py --block pt --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  bench --block pt: ms_per_step %.4f  regions %s' % (d['ms_per_step'], d.get('timed_regions_ms_per_step')))"
done
