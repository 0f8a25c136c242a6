export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; O=$R/gpurun_out/r06bench; mkdir -p $O
python -m pytest tests/test_gpu_pt_layer.py tests/test_gpu_blocks.py tests/test_gpu_model.py tests/test_gpu_bench_step_pt.py tests/test_gpu_bench_cli.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
python bench.py --block pt --steps 30 --warmup 5 > $O/bench_pt.json 2>$O/bench_pt.err
python - <<'P'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r06bench/bench_pt.json").read().strip().splitlines()[-1])
r=d["roofline"]; print(d["ms_per_step"], r["launch_us"], r.get("layer_bwd_us"), r["stage_ms"], d.get("no_pipeline",{}).get("ms_per_step"))
P
