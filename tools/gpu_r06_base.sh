#!/bin/bash
# round 6 baseline: the attention layer's kernels at both full-resolution shapes (rocprofv3 kernel stats)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06base; mkdir -p $O
export PYTHONPATH=$R
bash tools/gpu_prof_any.sh pt1664 45 python $R/tools/pt_layer_time.py 40960 16 64 > $O/pt1664.txt 2>&1
bash tools/gpu_prof_any.sh pt0832 45 python $R/tools/pt_layer_time.py 40960 8 32 > $O/pt0832.txt 2>&1
