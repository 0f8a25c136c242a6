#!/bin/bash
set -u
mkdir -p gpurun_out/r03d
O=gpurun_out/r03d
timeout 600 python -m pytest tests/test_gpu_local_aggregation.py tests/test_gpu_bench_convnet.py -q -x --timeout=300 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
timeout 300 python bench.py --workload convnet --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_convnet.json 2> $O/bench_convnet.err; echo "rc=$?"; python -c "
import json; d=json.load(open('$O/bench_convnet.json')); print(d['ms_per_step'], d['roofline']['stage_ms'], d['roofline'].get('adaptive_weight_bwd'))"
