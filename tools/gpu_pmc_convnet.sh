#!/bin/bash
# per-kernel counters of the ConvNet workload: separate --pmc passes with --kernel-trace only (no other trace domain), merged into one JSON
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmcconv
mkdir -p $O
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --workload convnet --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/p$i.log 2>&1; echo "pass $i ($set) rc=$?")
done
python - "$O" "${1:-unknown}" <<'PY'
import collections, csv, glob, json, re, sys
O, commit = sys.argv[1], sys.argv[2]
def short(name):
    n = re.sub(r"\(anonymous namespace\)::", "", name); n = re.sub(r"^void ", "", n)
    depth, out = 0, []
    for ch in n:
        if ch == "<": depth += 1
        elif ch == ">": depth -= 1
        elif ch == "(" and depth == 0: break
        out.append(ch)
    return "".join(out).strip()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
doc = {"_meta": {"command": "rocprofv3 --kernel-trace --pmc <set> -- python bench.py --workload convnet --steps 3 --warmup 1 --no-cpu-baseline --no-extra, five separate passes "
                            "(tools/gpu_pmc_convnet.sh)", "values": "mean per launch over all launches of the kernel (all five layers of the pyramid mixed)",
                 "units": "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them (gfx950: x2 on FETCH_SIZE for 16 B/lane reads, MI355X_MICROARCH.md); SQ_* summed over the device",
                 "commit": commit}}
for k, v in acc.items():
    doc[k] = {c: sum(x) / len(x) for c, x in v.items()}
    doc[k]["launches"] = max(len(x) for x in v.values())
json.dump(doc, open(O + "/pmc_convnet.json", "w"), indent=1)
for k in ("adaptive_weight_fwd_v5<3, 2>", "aw_bwd_csr_kernel<true, true, 2, 2>", "radius_group_kernel<32>", "radius_group_kernel<64>"):
    print(k, json.dumps(doc.get(k)))
PY
rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/p5
