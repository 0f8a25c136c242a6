#!/bin/bash
# full GPU suite (not -x: see everything a build-flag change breaks) + the three bench workloads
set -u
mkdir -p gpurun_out/r03k
export TMPDIR=/tmp
O=gpurun_out/r03k
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -30
timeout 120 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 120 python bench.py --block pt > $O/bench_pt.json 2> $O/bench_pt.err; echo "bench pt rc=$?"
timeout 200 python bench.py --workload convnet > $O/bench_convnet.json 2> $O/bench_convnet.err; echo "bench convnet rc=$?"
python - <<'PY'
import json
for f in ("bench", "bench_pt", "bench_convnet"):
    try:
        d = json.loads(open("gpurun_out/r03k/%s.json" % f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 4), d["value"], json.dumps(d["roofline"].get("stage_ms")))
        print("   ", {k: d["roofline"].get(k) for k in ("launch_us", "frac")}, json.dumps(d["roofline"].get("mfma_kpconv", d["roofline"].get("adaptive_weight")))[:300])
    except Exception as e:
        print(f, "unreadable", e)
PY
