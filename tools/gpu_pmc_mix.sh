#!/bin/bash
# per-kernel counters of a bench workload: separate --pmc passes with --kernel-trace only (no other trace domain), merged into one JSON
# usage: bash tools/gpu_pmc_mix.sh <commit> <tag> <bench.py arguments...>     e.g.  ... abc1234 convnet --workload convnet
set -u
export TMPDIR=/tmp
COMMIT=${1:-unknown}; TAG=${2:-bench}; shift 2 || true
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $O
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-pipeline > $O/p$i.log 2>&1; echo "pass $i ($set) rc=$?")
done
python - "$O" "$COMMIT" "$TAG $*" <<'PY'
import collections, csv, glob, json, re, sys
O, commit, what = sys.argv[1], sys.argv[2], sys.argv[3]
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
doc = {"_meta": {"command": "rocprofv3 --kernel-trace --pmc <set> -- python bench.py <args> --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-pipeline, five separate passes "
                            "(tools/gpu_pmc_mix.sh)", "bench_args": what, "values": "mean per launch over all launches of the kernel",
                 "units": "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them (gfx950: x2 on FETCH_SIZE for 16 B/lane reads, MI355X_MICROARCH.md); SQ_* summed over the device",
                 "commit": commit}}
for k, v in acc.items():
    doc[k] = {c: sum(x) / len(x) for c, x in v.items()}
    doc[k]["launches"] = max(len(x) for x in v.values())
for k, v in doc.items():
    if k == "_meta" or not v.get("GRBM_GUI_ACTIVE"): continue
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0
    v["duration_us_at_2.4GHz"] = round(cyc / 2400.0, 1)
    v["valu_issue_utilisation"] = round(v.get("SQ_INSTS_VALU", 0) * 4.0 / (cyc * 1024.0), 3)
    if v.get("SQ_WAVE_CYCLES"): v["wait_any_share"] = round(v.get("SQ_WAIT_ANY", 0) / v["SQ_WAVE_CYCLES"], 3)
doc["_meta"]["derived"] = ("valu_issue_utilisation = SQ_INSTS_VALU x 4 clk / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); wait_any_share = SQ_WAIT_ANY / SQ_WAVE_CYCLES; "
                           "duration = GRBM_GUI_ACTIVE / 8 / 2.4 GHz")
json.dump(doc, open(O + "/pmc_mix.json", "w"), indent=1)
top = sorted((k for k in doc if k != "_meta" and doc[k].get("GRBM_GUI_ACTIVE")), key=lambda k: -doc[k]["GRBM_GUI_ACTIVE"] * doc[k]["launches"])[:14]
for k in top:
    print(k[:58].ljust(58), doc[k]["launches"], doc[k]["duration_us_at_2.4GHz"], doc[k]["valu_issue_utilisation"], doc[k].get("wait_any_share"), int(doc[k].get("SQ_INSTS_VALU", 0)))
PY
rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/p5
