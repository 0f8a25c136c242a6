#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -6
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_h.json 2> gpurun_out/bench_h.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_h.json"))
print("ms/step", d["ms_per_step"], "value", d["value"], "fwd-only ms", d["forward_only"]["ms_per_step"])
print(json.dumps(d["roofline"]["stage_ms"]))
PY
