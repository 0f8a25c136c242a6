#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
for i in 1 2 3 4; do
  O=$GRAFT_REPO_ROOT/gpurun_out/ptrace_$i; rm -rf $O; mkdir -p $O
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/raw -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --no-legs --no-gather-200k --steps 120 --warmup 10 > $O/run.log 2>&1)
  grep -o '"ms_per_step": [0-9.]*' $O/run.log | head -1
  f=$(find $O/raw -name "*kernel_trace.csv" | head -1); python tools/pipeline_trace.py $f; rm -rf $O/raw
done
