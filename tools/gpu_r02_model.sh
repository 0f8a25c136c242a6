#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
timeout 600 python tools/bench_model.py --graph 2>&1 | tail -5
bash tools/exp/prof_one.sh model $GRAFT_REPO_ROOT/tools/bench_model.py --graph > /dev/null
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_model/p_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6, "kernels", sum(int(r["Calls"]) for r in rows))
for r in rows[:45]:
    name=r["Name"].replace("(anonymous namespace)::","").replace("void ","")
    print(f'{name[:70]:70s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:8.2f} pct={float(r["Percentage"]):5.1f}')
PY
