#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_blocks.py -x -q 2>&1 | tail -3
bash tools/gpu_r02_model.sh
