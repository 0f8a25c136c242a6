#!/bin/bash
# same-box A/B of the fused attention layer's kernels across builds of the library: bash tools/gpu_pt_ab.sh <lib.so | ""> ...   ("" = the in-tree build)
# per build: rocprofv3 kernel averages of tools/pt_layer_time.py (40960, 16, 64) and (40960, 8, 32) side by side, then the pipelined --block pt step
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/pt_ab
mkdir -p $O
i=0
for lib in "$@"; do
  i=$((i+1))
  tag=$(basename "${lib:-intree}" .so)${CBL_PT_NARROW_ROWS:+_rows$CBL_PT_NARROW_ROWS}
  for shape in "40960 16 64" "40960 8 32"; do
    s=$(echo $shape | tr ' ' '_')
    (cd /tmp && CBL_AMD_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o p -- python $GRAFT_REPO_ROOT/tools/pt_layer_time.py $shape > $O/run_${tag}_$s.log 2>&1)
    f=$(find $O/raw -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${i}_${tag}_$s.csv; rm -rf $O/raw
  done
  CBL_AMD_LIB=$lib timeout 300 python bench.py --block pt --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench --block pt [%s]: ms_per_step %.4f  regions %s' % ('$tag', d['ms_per_step'], d.get('timed_regions_ms_per_step')))"
done
python - "$O" <<'PY'
import csv, glob, os, re, sys
O = sys.argv[1]
files = sorted(glob.glob(O + "/*.csv"))
table, cols = {}, []
for f in files:
    col = os.path.basename(f)[:-4]; cols.append(col)
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"^void ", "", n); n = n.split("(")[0]
        if n.startswith(("pt_", "triple")):
            table.setdefault(n, {})[col] = float(r["AverageNs"]) / 1e3
for shape in ("40960_16_64", "40960_8_32"):
    cs = [c for c in cols if c.endswith(shape)]
    print("== %s   columns: %s" % (shape, "  |  ".join(c[:-len(shape) - 1] for c in cs)))
    rows = [(n, [v.get(c) for c in cs]) for n, v in table.items() if any(c in v for c in cs)]
    for n, vals in sorted(rows, key=lambda t: -(t[1][0] or 0)):
        print("  %-40s %s" % (n[:40], "  ".join("%7.1f" % x if x is not None else "      -" for x in vals)))
    print("  %-40s %s" % ("SUM", "  ".join("%7.1f" % sum(v.get(c, 0) for v in table.values()) for c in cs)))
PY
