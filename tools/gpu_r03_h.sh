#!/bin/bash
set -u
mkdir -p gpurun_out/r03h; ulimit -c 0
O=gpurun_out/r03h
timeout 600 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_bench_step_pt.py tests/test_gpu_model.py -q -x --timeout=400 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -v "^$" $O/pytest.log | grep "passed\|failed\|Error\|worst\|rc=" | head
timeout 400 python bench.py --block pt --steps 20 --warmup 3 > $O/bench_pt.json 2> $O/bench_pt.err; echo "rc=$?"; python -c "
import json; d=json.load(open('$O/bench_pt.json')); print(d['ms_per_step'], d['roofline']['launch_us'], d['roofline']['stage_ms'], d['no_pipeline']['ms_per_step'])"
