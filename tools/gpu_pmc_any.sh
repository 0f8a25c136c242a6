#!/bin/bash
# per-kernel counters of ANY command: separate --pmc passes with --kernel-trace only, merged into gpurun_out/pmc_<tag>/pmc_mix.json
# usage: bash tools/gpu_pmc_any.sh <tag> <kernel-name regex to print> <command...>
set -u
export TMPDIR=/tmp
TAG=$1; PAT=$2; shift 2
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $O
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o pmc -- "$@" > $O/p$i.log 2>&1; echo "pass $i ($set) rc=$?")
done
python - "$O" "$TAG" "$PAT" <<'PY'
import collections, csv, glob, json, re, sys
O, tag, pat = sys.argv[1], sys.argv[2], sys.argv[3]
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
doc = {"_meta": {"command": "rocprofv3 --kernel-trace --pmc <set> -- <command>, separate passes (tools/gpu_pmc_any.sh)", "tag": tag, "values": "mean per launch",
                 "units": "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them (gfx950: x2 on FETCH_SIZE for wide reads, MI355X_MICROARCH.md); SQ_* summed over the device"}}
for k, v in acc.items():
    doc[k] = {c: sum(x) / len(x) for c, x in v.items()}
    doc[k]["launches"] = max(len(x) for x in v.values())
for k, v in doc.items():
    if k == "_meta" or not v.get("GRBM_GUI_ACTIVE"): continue
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0
    v["duration_us_at_2.4GHz"] = round(cyc / 2400.0, 1)
    v["valu_issue_utilisation"] = round(v.get("SQ_INSTS_VALU", 0) * 4.0 / (cyc * 1024.0), 3)
    if v.get("SQ_WAVE_CYCLES"): v["wait_any_share"] = round(v.get("SQ_WAIT_ANY", 0) / v["SQ_WAVE_CYCLES"], 3)
    if v.get("TCC_HIT_sum") is not None: v["l2_hit_rate"] = round(v["TCC_HIT_sum"] / max(v["TCC_HIT_sum"] + v.get("TCC_MISS_sum", 0), 1), 3)
json.dump(doc, open(O + "/pmc_mix.json", "w"), indent=1)
rx = re.compile(pat)
top = sorted((k for k in doc if k != "_meta" and doc[k].get("GRBM_GUI_ACTIVE") and rx.search(k)), key=lambda k: -doc[k]["GRBM_GUI_ACTIVE"])[:24]
print("%-44s %4s %7s %6s %6s %6s %9s %9s %9s" % ("kernel", "n", "us", "valu", "wait", "l2hit", "fetchMB*2", "writeMB", "instVALU"))
for k in top:
    v = doc[k]
    print("%-44s %4d %7.1f %6.3f %6s %6s %9.1f %9.1f %9d" % (k[:44], v["launches"], v["duration_us_at_2.4GHz"], v["valu_issue_utilisation"], v.get("wait_any_share"), v.get("l2_hit_rate"),
          2 * v.get("FETCH_SIZE", 0) / 1024, v.get("WRITE_SIZE", 0) / 1024, int(v.get("SQ_INSTS_VALU", 0))))
PY
rm -rf $O/p[0-9]*
