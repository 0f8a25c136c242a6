#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do for v in 100 75 50 35 150; do
CBL_QG_PCT=$v timeout 300 python bench.py --no-cpu-baseline --no-legs --no-gather-200k --steps 200 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('pct=$v ms_per_step %.4f no_pipeline %.4f gather us %.2f frac %.3f stage %.4f' % (d['ms_per_step'], d['no_pipeline']['ms_per_step'], r['launch_us'], r['frac'], r['stage_ms']['queryandgroup']))"
done; done
