#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for g in 512 768 1024 1536; do CBL_KB_GRID=$g timeout 120 python tools/exp/kb_probe.py 2>&1 | grep -v amdgpu.ids | head -1; done
timeout 300 python -m pytest tests/test_gpu_local_aggregation.py tests/test_gpu_bench_step.py -m gpu -q -x 2>&1 | tail -2
