#!/bin/bash
# every schedule flag of bench.py once (short runs), and __graft_entry__.smoke()
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for f in "" "--no-pipeline" "--no-graph" "--no-overlap" "--no-nested" "--forward-only" "--forward-only --no-pipeline"; do
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra $f 2>gpurun_out/err_flags.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],4), d['config']['issue'][:60])" || tail -3 gpurun_out/err_flags.txt
done
