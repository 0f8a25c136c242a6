#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_pointops.py tests/test_gpu_bench_step.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -2
timeout 300 python bench.py --no-cpu-baseline --no-legs --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms_per_step %.4f no_pipeline %.4f' % (d['ms_per_step'], d['no_pipeline']['ms_per_step']))
print('gather us %.2f frac %.3f of_fill %.3f' % (r['launch_us'], r['frac'], r['frac_of_measured_fill']))
g=r['gather_200k']; print('gather_200k us %.2f frac %.3f of_fill %s' % (g['launch_us'], g['frac'], g.get('frac_of_fill')))
print('stage', r['stage_ms']['queryandgroup'])"
