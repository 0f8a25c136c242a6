#!/bin/bash
# usage: pmc_insts.sh <tag> <python args...>: SQ instruction-mix counters per kernel (kernel-trace + pmc only)
tag=$1; shift
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmci_$tag/$n -o p -- python "$@" > /dev/null 2>&1)
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmci_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== $tag")
for k, v in acc.items():
    if "$tag" != "all" and "knn" not in k: continue
    print(k)
    print("   ", {c: round(sum(x) / len(x)) for c, x in sorted(v.items())})
PY
