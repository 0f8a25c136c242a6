#!/bin/bash
# network step after a change: the block / model tests, then the graph step (1 and 4 scenes) and the flat-state data-parallel step on a one-rank group
set -u
out=gpurun_out/model_ab; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_model.py tests/test_gpu_pt_layer.py -q -x -m gpu 2>&1 | tail -4
model() { tag=$1; shift; timeout 300 python tools/bench_model.py "$@" 2> $out/model_$tag.err | grep '^{' | tail -1 > $out/model_$tag.json; python - $out/model_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print(sys.argv[2], "ms_per_step", round(d["ms_per_step"], 3), round(d["points_per_s"] / 1e6, 3), "M pts/s", "replay", (d.get("segments_ms") or {}).get("replay_ms"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
model graph --graph --steps 10 --warmup 3
model graph_4scenes --graph --scenes 4 --steps 6 --warmup 2
model srg_flat --single-rank-group --graph --steps 10 --warmup 3
for a in "$@"; do eval "$a"; done
