timeout 150 python -m pytest tests/test_gpu_nested.py tests/test_gpu_hotpath.py tests/test_gpu_order.py -x -q 2>&1 | tail -3
for i in 1 2 3; do
timeout 80 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(round(d['ms_per_step'],4),d['roofline']['stage_ms'])"
done
