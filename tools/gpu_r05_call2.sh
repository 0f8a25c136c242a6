#!/bin/bash
# round 5, second GPU call: the apply pass with d p1 on the matrix cores against round 4's library on the same box; the narrow passes' grid; where the
# data-parallel graph step's missing milliseconds go (geometry streams probed / unprobed, host clocks per segment)
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05c2
mkdir -p $O
cd $GRAFT_REPO_ROOT
R04=$GRAFT_REPO_ROOT/contrastboundary_amd/lib/libcbl_amd_r04.so
timeout 900 python -m pytest tests/test_gpu_pt_layer.py tests/test_gpu_blocks.py tests/test_gpu_bench_step_pt.py -q -x --timeout=600 -m gpu > $O/pytest_pt.log 2>&1; echo "pytest rc=$?" >> $O/pytest_pt.log
tail -4 $O/pytest_pt.log
for rep in 1 2; do
  for lib in "" "$R04"; do
    echo "== pt_layer_time lib=${lib:-new} rep=$rep"
    CBL_AMD_LIB=$lib timeout 200 python tools/pt_layer_time.py 40960 16 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['new'])"
  done
done
CBL_AMD_LIB= timeout 200 python tools/pt_layer_time.py 40960 8 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new 8/32', d['new'])"
CBL_AMD_LIB=$R04 timeout 200 python tools/pt_layer_time.py 40960 8 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r04 8/32', d['new'])"
for rows in 1024 2048; do
  echo "== narrow rows $rows"
  CBL_PT_NARROW_ROWS=$rows timeout 200 python tools/pt_layer_time.py 40960 16 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['new'])"
done
bash tools/gpu_prof_any.sh r05c2_pt 30 python $GRAFT_REPO_ROOT/tools/pt_layer_time.py 40960 16 64 | grep -E "pt_|triple|total"
CBL_PT_NARROW_ROWS=2048 bash tools/gpu_prof_any.sh r05c2_pt_rows2048 30 python $GRAFT_REPO_ROOT/tools/pt_layer_time.py 40960 16 64 | grep -E "pt_pchain|pt_narrow|pt_bn|pt_softmax|pt_sum"
# pipelined block step, new against round 4's library
for lib in "" "$R04" ""; do
  CBL_AMD_LIB=$lib timeout 300 python bench.py --block pt --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench --block pt lib=%s ms_per_step %.4f' % ('${lib:-new}'[-12:], d['ms_per_step']))"
done
# data-parallel graph step
run_model() { tag=$1; shift; timeout 300 python tools/bench_model.py "$@" > $O/model_$tag.json 2> $O/model_$tag.err; echo "model $tag rc=$?"; grep '^{' $O/model_$tag.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ms_per_step %.2f' % d['ms_per_step'], json.dumps(d.get('segments_ms')))"; }
run_model graph --graph --steps 10 --warmup 3
run_model srg_flat --single-rank-group --graph --steps 10 --warmup 3
CBL_GEO_STREAMS=unprobed_first run_model srg_flat_unprobed --single-rank-group --graph --steps 10 --warmup 3
run_model srg_flat_nobcast --single-rank-group --graph --no-buffer-broadcast --steps 10 --warmup 3
run_model srg_flat_d3 --single-rank-group --graph --depth 3 --steps 10 --warmup 3
run_model graph_4scenes --graph --scenes 4 --steps 6 --warmup 2
run_model srg_flat_4scenes --single-rank-group --graph --scenes 4 --steps 6 --warmup 2
