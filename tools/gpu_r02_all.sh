#!/bin/bash
# round 2, everything that is recorded: the whole -m gpu suite, the final bench / rocprof / PMC set, the full-network step
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|Error" | tail -4
bash tools/gpu_r02_final.sh
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
(timeout 600 python tools/bench_model.py --graph 2>/dev/null | tail -1; timeout 600 python tools/bench_model.py --graph --depth 1 2>/dev/null | tail -1; timeout 600 python tools/bench_model.py --graph --scenes 4 2>/dev/null | tail -1) > gpurun_out/r02_bench_model.jsonl
cut -c1-140 gpurun_out/r02_bench_model.jsonl
