export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; O=$R/gpurun_out/r06bench; mkdir -p $O
python -m pytest tests/test_gpu_pointops.py tests/test_gpu_hotpath.py tests/test_gpu_bench_step.py tests/test_gpu_order.py tests/test_gpu_nested.py -x -q -m gpu > $O/tests2.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests2.log
python bench.py --no-legs --no-cpu-baseline --steps 30 > $O/bench_head.json 2>$O/bench_head.err
python - <<'P'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r06bench/bench_head.json").read().strip().splitlines()[-1])
r=d["roofline"]; print(d["ms_per_step"], d.get("no_pipeline",{}).get("ms_per_step"), r["stage_ms"])
P
