#!/bin/bash
# per-dispatch kernel trace of ANY command: bash tools/gpu_trace_any.sh <tag> <command...>  -> gpurun_out/trace_<tag>/kernel_trace.csv (name, start, end, grid, workgroup)
set -u
export TMPDIR=/tmp
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG
mkdir -p $O
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/raw -o t -- "$@" > $O/run.log 2>&1; echo "rocprof rc=$?")
f=$(find $O/raw -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 - "$f" "$O/kernel_trace.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Workgroup_Size_X", "Stream_Id", "Queue_Id"]
keep = [k for k in keep if rows and k in rows[0]]
w = csv.writer(open(sys.argv[2], "w")); w.writerow(keep)
for r in rows:
    w.writerow([r[k][:100] if k == "Kernel_Name" else r[k] for k in keep])
print("dispatches", len(rows), "columns", list(rows[0].keys()) if rows else None)
PY
rm -rf $O/raw
ls -la $O
