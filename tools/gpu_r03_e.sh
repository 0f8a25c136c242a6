#!/bin/bash
set -u
mkdir -p gpurun_out/r03e
O=gpurun_out/r03e
timeout 900 python -m pytest tests/test_gpu_cbl.py tests/test_gpu_local_aggregation.py tests/test_gpu_transpose.py tests/test_gpu_model.py tests/test_gpu_bench_step.py tests/test_gpu_bench_convnet.py -q --timeout=600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -v "^$" $O/pytest.log | tail -40
timeout 200 python __graft_entry__.py --smoke 2>&1 | tail -2
