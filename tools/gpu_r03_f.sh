#!/bin/bash
set -u
mkdir -p gpurun_out/r03f
O=gpurun_out/r03f
timeout 600 python -m pytest tests/test_gpu_bench_step_pt.py -q -x --timeout=400 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -v "^$" $O/pytest.log | tail -25
timeout 400 python bench.py --block pt --steps 20 --warmup 3 > $O/bench_pt.json 2> $O/bench_pt.err; echo "rc=$?"; cut -c1-2500 $O/bench_pt.json; tail -5 $O/bench_pt.err
