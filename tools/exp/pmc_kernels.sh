#!/bin/bash
# usage: pmc_kernels.sh <tag> <filter-regex> <python args...>: SQ busy / wait / instruction counters per kernel (kernel-trace + pmc only, separate passes)
tag=$1; shift; filt=$1; shift
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmck_$tag/$n -o p -- python "$@" > /dev/null 2>&1)
done
python - <<PY
import csv, glob, collections, re, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmck_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, v in acc.items():
    if not re.search(r"$filt", k): continue
    out[k] = {c: round(sum(x) / len(x)) for c, x in sorted(v.items())}
    print(k); print("   ", out[k])
json.dump(out, open("$GRAFT_REPO_ROOT/gpurun_out/pmck_$tag.json", "w"), indent=1)
PY
