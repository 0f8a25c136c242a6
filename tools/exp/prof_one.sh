#!/bin/bash
# usage: prof_one.sh <tag> <python args...>  -> gpurun_out/prof_<tag>/..._kernel_stats.csv (printed)
tag=$1; shift
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o p -- python "$@" > /dev/null 2>&1)
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
echo "== $tag"; cut -d, -f1-4 "$f" | head -14
