#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_transpose.py tests/test_gpu_pointops.py tests/test_gpu_blocks.py tests/test_gpu_model.py -x -q 2>&1 | tail -12
