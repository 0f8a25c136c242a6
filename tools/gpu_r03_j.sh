#!/bin/bash
set -u
mkdir -p gpurun_out/r03j
export TMPDIR=/tmp
O=gpurun_out/r03j
timeout 600 python -m pytest tests/test_gpu_local_aggregation.py -m gpu -q -x --timeout=300 -k "adaptive" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -6
for cfg in "3 2" "3 1" "3 4" "2 2"; do
set -- $cfg
CBL_AW_CPL=$1 CBL_AW_UB=$2 timeout 200 python bench.py --workload convnet > $O/bench_convnet_c$1_b$2.json 2> $O/bench_convnet_c$1_b$2.err; echo "bench convnet CPL=$1 UB=$2 rc=$?"
done
python - <<'PY'
import json
for f in ("c3_b2", "c3_b1", "c3_b4", "c2_b2"):
    try:
        d = json.loads(open("gpurun_out/r03j/bench_convnet_%s.json" % f).read().strip().splitlines()[-1])
        st = d["roofline"]["stage_ms"]
        print(f, round(d["ms_per_step"], 3), [st["adaptive_weight_bwd_l%d" % l] for l in range(5)], d["roofline"]["adaptive_weight"]["launch_us"], d["roofline"]["adaptive_weight_bwd"]["launch_us"])
    except Exception as e:
        print(f, "unreadable", e)
PY
