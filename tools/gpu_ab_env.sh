#!/bin/bash
# A/B/... of one environment switch inside ONE gpurun call (box-to-box spread is larger than most differences): usage
#   bash tools/gpu_ab_env.sh <out dir under gpurun_out> <VAR> <value,value,...> -- <bench.py arguments>
# runs `python bench.py <arguments>` with VAR set to every value in turn, three rounds, and prints ms_per_step / no_pipeline per run
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/$1; VAR=$2; VALS=$3; shift 4
mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in ${VALS//,/ }; do
    env $VAR=$v timeout 300 python bench.py "$@" > $O/run_${v}_$rep.json 2> $O/run_${v}_$rep.err
    python - "$O/run_${v}_$rep.json" "$VAR=$v" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[2], "ms_per_step %.4f" % d["ms_per_step"], "no_pipeline %s" % (d.get("no_pipeline") or {}).get("ms_per_step"), "regions", d.get("timed_regions_ms_per_step"))
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
  done
done
