#!/bin/bash
# round 4, one GPU-box call for the Point-Transformer layer work: parity tests of the new path, timings, kernel stats.  Outputs: gpurun_out/r04pt/
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04pt
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pt_layer.py tests/test_gpu_dense.py tests/test_gpu_blocks.py tests/test_gpu_bench_step_pt.py -q -x -s --timeout=600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -30
timeout 300 python tools/pt_layer_time.py 40960 16 64 > $O/time_40960_16_64.json 2> $O/time_a.err; cat $O/time_40960_16_64.json; tail -3 $O/time_a.err
timeout 300 python tools/pt_layer_time.py 40960 8 32 > $O/time_40960_8_32.json 2> $O/time_b.err; cat $O/time_40960_8_32.json; tail -3 $O/time_b.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o pt -- python $GRAFT_REPO_ROOT/tools/pt_layer_time.py 40960 16 64 > $O/prof.log 2>&1; echo "rocprof rc=$?")
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/pt_layer_kernel_stats.csv && rm -rf $O/prof
python - <<'PY'
import csv, os, re
f = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r04pt/pt_layer_kernel_stats.csv")
if os.path.exists(f):
    for r in list(csv.DictReader(open(f)))[:40]:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"^void ", "", n); n = n.split("(")[0][:60]
        print("%-60s calls %5s avg_us %8.1f pct %s" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
timeout 400 python bench.py --block pt --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_pt.json 2> $O/bench_pt.err; echo "bench pt rc=$?"; cut -c1-600 $O/bench_pt.json; tail -3 $O/bench_pt.err
