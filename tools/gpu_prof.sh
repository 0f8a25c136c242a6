#!/bin/bash
# rocprofv3 kernel stats of one bench.py command line:  bash tools/gpu_prof.sh <tag> <bench.py args...>
set -u
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/prof_$tag
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o $tag -- python $GRAFT_REPO_ROOT/bench.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag/run.log 2>&1; echo "rocprof rc=$?")
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -24 "$f" | cut -c1-110,300-400
