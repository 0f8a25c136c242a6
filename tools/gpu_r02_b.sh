#!/bin/bash
# round 2, call B: probes with per-kernel rocprof, new tests, bench
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
timeout 600 python tools/exp/r02_probe.py > gpurun_out/r02_probe_b.log 2>&1; echo "probe rc=$?" >> gpurun_out/r02_probe_b.log; cat gpurun_out/r02_probe_b.log
timeout 900 python -m pytest tests/test_gpu_transpose.py tests/test_gpu_bench_step.py tests/test_gpu_dropin.py tests/test_gpu_local_aggregation.py tests/test_gpu_hotpath.py tests/test_gpu_nested.py -m gpu -q -x --timeout=600 > gpurun_out/pytest_new.log 2>&1; echo "pytest-new rc=$?" >> gpurun_out/pytest_new.log
tail -30 gpurun_out/pytest_new.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err; echo "bench rc=$?"; cat gpurun_out/bench_b.json; tail -5 gpurun_out/bench_b.err
bash tools/exp/prof_one.sh r02b $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --steps 20 | cut -c1-160
f=$(find gpurun_out/prof_r02b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-5 "$f" | head -40
bash tools/exp/prof_one.sh r02probe $GRAFT_REPO_ROOT/tools/exp/r02_probe.py | cut -c1-160
f=$(find gpurun_out/prof_r02probe -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-5 "$f" | grep -i "nt_\|csr\|contrast\|kpconv" | head -30
