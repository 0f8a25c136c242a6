#!/bin/bash
set -u
mkdir -p gpurun_out/r03l
export TMPDIR=/tmp
O=gpurun_out/r03l
timeout 900 python -m pytest tests/test_gpu_tfops.py tests/test_gpu_bench_convnet.py -m gpu -q --timeout=600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -8
timeout 200 python bench.py --workload convnet > $O/bench_convnet.json 2> $O/bench_convnet.err; echo "bench convnet rc=$?"
python - <<'PY'
import json
for f in ("bench_convnet",):
    try:
        d = json.loads(open("gpurun_out/r03l/%s.json" % f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 4), d["value"], json.dumps(d["roofline"].get("stage_ms")))
        print(d["roofline"]["launch_us"], d["roofline"]["frac"], json.dumps(d["cpu_baseline"])[:400])
    except Exception as e:
        print(f, "unreadable", e)
PY
