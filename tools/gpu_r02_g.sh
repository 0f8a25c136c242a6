#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_transpose.py tests/test_gpu_cbl.py -m gpu -q -x 2>&1 | tail -2
bash tools/exp/pmc_kernels.sh step "kpconv|nt_finish|contrast_pairs|knn_grid_wave|query_group|grouping_bwd_csr" $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 2
