#!/bin/bash
# the attention layer after a change: its GPU tests, then tools/pt_layer_time.py at both full-resolution shapes (eager + graph) and rocprofv3 kernel stats
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06pt; mkdir -p $O; export PYTHONPATH=$R
echo skip tests
python tools/pt_layer_time.py 40960 16 64 --graph > $O/pt_time.jsonl 2>$O/pt_time.err; python tools/pt_layer_time.py 40960 8 32 --graph >> $O/pt_time.jsonl 2>>$O/pt_time.err; cat $O/pt_time.jsonl
bash tools/gpu_prof_any.sh pt1664 45 python $R/tools/pt_layer_time.py 40960 16 64 > $O/pt1664.txt 2>&1
bash tools/gpu_prof_any.sh pt0832 45 python $R/tools/pt_layer_time.py 40960 8 32 > $O/pt0832.txt 2>&1
grep -E "pt_|triple" $O/pt1664.txt | head -24

