#!/bin/bash
# round 4: whole GPU suite + the three bench lines (outputs: gpurun_out/r04check/)
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04check
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -x --timeout=900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | grep -v Warning | tail -15
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 600 python bench.py --workload convnet --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_convnet.json 2> $O/bench_convnet.err; echo "bench convnet rc=$?"
python - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r04check")
try:
    d = json.load(open(O + "/bench_convnet.json"))
    print("convnet ms/step", round(d["ms_per_step"], 3), "host issue", round(d["host_issue_ms_per_step"], 3), d["roofline"]["stage_ms"])
except Exception as e:
    print("convnet parse failed", e)
PY
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r04check")
try:
    d = json.load(open(O + "/bench.json"))
    r = d["roofline"]
    print("headline ms/step", round(d["ms_per_step"], 4), "no_pipeline", round(d.get("no_pipeline", {}).get("ms_per_step", 0), 4), "gather frac", round(r["frac"], 3), "us", r["launch_us"])
    print("gather_200k", {k: r.get("gather_200k", {}).get(k) for k in ("frac", "launch_us", "fill_same_size_GBps", "frac_of_fill", "error")})
    print("stage_ms", r["stage_ms"])
    for k in ("pt_block", "convnet"):
        v = d.get(k, {})
        print(k, {a: v.get(a) for a in ("ms_per_step", "value", "error")}, (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("frac_of_f32_mfma_peak"))
    print("cpu_baseline", {k: d.get("cpu_baseline", {}).get(k) for k in ("value", "unit", "cores", "kind")})
except Exception as e:
    print("bench parse failed", e)
PY
tail -3 $O/bench.err
