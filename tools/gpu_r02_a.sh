#!/bin/bash
# round 2, call A: probes + the new tests first, then the whole GPU suite, smoke, bench
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
python -c "import torch; print(torch.cuda.get_device_name(0))" > gpurun_out/device.txt 2>&1
timeout 600 python tools/exp/r02_probe.py > gpurun_out/r02_probe.log 2>&1; echo "probe rc=$?" >> gpurun_out/r02_probe.log; cat gpurun_out/r02_probe.log
timeout 900 python -m pytest tests/test_gpu_transpose.py tests/test_gpu_bench_step.py tests/test_gpu_dropin.py tests/test_gpu_cbl.py -m gpu -q --timeout=600 > gpurun_out/pytest_new.log 2>&1; echo "pytest-new rc=$?" >> gpurun_out/pytest_new.log
tail -40 gpurun_out/pytest_new.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err; echo "bench rc=$?"; cat gpurun_out/bench_a.json; tail -5 gpurun_out/bench_a.err
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 --deselect tests/test_gpu_transpose.py --deselect tests/test_gpu_bench_step.py > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
