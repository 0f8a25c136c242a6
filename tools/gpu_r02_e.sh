#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -4
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_e.json 2> gpurun_out/bench_e.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_e.json"))
print("ms/step", d["ms_per_step"], "value", d["value"], "fwd-only ms", d["forward_only"]["ms_per_step"])
print(json.dumps(d["roofline"]["stage_ms"]))
PY
tail -3 gpurun_out/bench_e.err
bash tools/exp/prof_one.sh r02e $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --steps 20 > /dev/null
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_r02e/p_kernel_stats.csv")))
for r in rows[:24]:
    name=r["Name"].replace("(anonymous namespace)::","").replace("void ","")
    print(f'{name[:58]:58s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:8.2f} pct={float(r["Percentage"]):5.1f}')
PY
