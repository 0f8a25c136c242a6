#!/bin/bash
# round 3, second half: the new pieces (TF sample strings / 'S' margin, PosPool gather backward, pyramid with deferred syncs, index_max / param reduce)
set -u
mkdir -p gpurun_out/r03i
export TMPDIR=/tmp
O=gpurun_out/r03i
timeout 600 python -m pytest tests/test_gpu_cbl.py tests/test_gpu_local_aggregation.py tests/test_gpu_tfops.py tests/test_gpu_bench_convnet.py -m gpu -q -x --timeout=300 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -15
timeout 200 python bench.py --workload convnet > $O/bench_convnet.json 2> $O/bench_convnet.err; echo "bench convnet rc=$?"
timeout 120 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
for f in ("bench_convnet", "bench"):
    try:
        d = json.loads(open("gpurun_out/r03i/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], json.dumps(d["roofline"].get("stage_ms")))
    except Exception as e:
        print(f, "unreadable", e)
PY
