#!/bin/bash
# round 2: the numbers that go into profiles/ — PMC traffic first (separate passes; bench.py reads the summary for roofline.traffic), then the bench line,
# the rocprof kernel stats of the same command, the PMC instruction mix
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
python -c "import torch; print(torch.cuda.get_device_name(0))" > gpurun_out/device.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_pmc/$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/gpurun_out/r02_pmc_$c.log 2>&1; echo "$c rc=$?")
done
F=$(find gpurun_out/r02_pmc/FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/r02_pmc/WRITE_SIZE -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_summary.py "$F" "$W" gpurun_out/r02_pmc_traffic.json && cp gpurun_out/r02_pmc_traffic.json profiles/r02_pmc_traffic.json \
  && cp "$F" gpurun_out/r02_pmc_FETCH_SIZE_counter_collection.csv && cp "$W" gpurun_out/r02_pmc_WRITE_SIZE_counter_collection.csv
timeout 900 python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r02_bench.json; tail -3 gpurun_out/r02_bench.err
timeout 600 python bench.py --forward-only --no-cpu-baseline > gpurun_out/r02_bench_forward_only.json 2>/dev/null; echo "bench fwd rc=$?"
timeout 600 python bench.py --no-pipeline --no-cpu-baseline --no-extra > gpurun_out/r02_bench_no_pipeline.json 2>/dev/null; echo "bench no-pipeline rc=$?"; cut -c1-260 gpurun_out/r02_bench_no_pipeline.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r02_prof.log 2>&1; echo "rocprof rc=$?")
f=$(find gpurun_out/r02_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r02_bench_kernel_stats.csv && cut -d, -f1-4 "$f" | cut -c1-120 | head -16
rm -rf gpurun_out/r02_prof/*kernel_trace.csv
bash tools/exp/pmc_kernels.sh r02 "." $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 2 > gpurun_out/r02_pmc_kernels.log 2>&1; tail -2 gpurun_out/r02_pmc_kernels.log | cut -c1-200
