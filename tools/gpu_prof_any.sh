#!/bin/bash
# rocprofv3 kernel stats of ANY command: bash tools/gpu_prof_any.sh <tag> <rows> <command...>  -> gpurun_out/prof_<tag>/kernel_stats.csv + top rows
set -u
export TMPDIR=/tmp
TAG=$1; ROWS=$2; shift 2
O=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $O
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o p -- "$@" > $O/run.log 2>&1; echo "rocprof rc=$?")
f=$(find $O/raw -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv && rm -rf $O/raw
python3 - "$O/kernel_stats.csv" "$ROWS" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms %.2f" % (tot / 1e6))
for r in rows[:int(sys.argv[2])]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"^void ", "", n); n = n.split("(")[0][:58]
    print("%-58s calls %5s avg_us %8.1f tot_ms %7.2f pct %s" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
