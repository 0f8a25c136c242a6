#!/bin/bash
# round 5, first GPU call: the new tests, and where the data-parallel wrapper's time goes (old layout, new layout, no wrapper)
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05c1
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/device.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_pt_layer.py tests/test_gpu_model.py -q -x --timeout=600 -m gpu > $O/pytest_new.log 2>&1; echo "pytest rc=$?" >> $O/pytest_new.log
tail -15 $O/pytest_new.log
timeout 300 python tools/bench_model.py --graph --steps 10 --warmup 3 > $O/model_graph.json 2> $O/model_graph.err; echo "model graph rc=$?"
timeout 300 python tools/bench_model.py --single-rank-group --graph --steps 10 --warmup 3 > $O/model_graph_srg_flat.json 2> $O/model_graph_srg_flat.err; echo "model srg flat rc=$?"
timeout 300 python tools/bench_model.py --single-rank-group --graph --hook-reducer --steps 10 --warmup 3 > $O/model_graph_srg_hook.json 2> $O/model_graph_srg_hook.err; echo "model srg hook rc=$?"
timeout 300 python tools/bench_model.py --single-rank-group --steps 10 --warmup 3 > $O/model_eager_srg_flat.json 2> $O/model_eager_srg_flat.err; echo "model eager srg flat rc=$?"
timeout 300 python tools/pt_layer_time.py 40960 16 64 > $O/pt_layer_40960_16_64.json 2>/dev/null; echo "pt layer rc=$?"
cat $O/model_*.json | cut -c1-900
tail -3 $O/*.err | tail -40
