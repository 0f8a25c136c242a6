#!/bin/bash
# HBM traffic counters of the bench kernels: separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc/$c.log 2>&1; echo "$c rc=$?")
done
find gpurun_out/pmc -name "*.csv" | head
python - <<'PY'
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"gpurun_out/pmc/{c}/**/*counter_collection.csv", recursive=True)
    if not files: print(c, "no counter csv"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r.get("Counter_Name") == c:
            acc[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
    print("==", c, "(KB per launch, mean over launches)")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]) / len(kv[1]))[:12]:
        print(f"  {k:70s} n={len(v):3d} mean={sum(v)/len(v):12.1f}")
PY
