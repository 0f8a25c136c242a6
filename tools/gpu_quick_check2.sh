#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 200 python tools/fwd_determinism.py 2>&1 | grep -v Warn | grep "^run" | cut -c1-120
timeout 600 python -m pytest tests/test_gpu_pointops.py tests/test_gpu_nested.py tests/test_gpu_hotpath.py tests/test_gpu_bench_step.py -x -q 2>&1 | tail -2
bash tools/gpu_prof_any.sh hl 26 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --no-legs --no-gather-200k --no-overlap --no-pipeline --steps 100 --warmup 5 | grep "canon\|wave\|replay\|total"
timeout 300 python bench.py --no-cpu-baseline --no-legs --no-gather-200k --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step %.4f no_pipeline %.4f knn stage %.4f' % (d['ms_per_step'], d['no_pipeline']['ms_per_step'], d['roofline']['stage_ms']['knnquery_k16']))"
