#!/bin/bash
# round 3, call A: the new ConvNet workload + rank-aware training harness on the GPU box
set -u
mkdir -p gpurun_out/r03a
export TMPDIR=/tmp
O=gpurun_out/r03a
timeout 900 python -m pytest tests/test_gpu_bench_convnet.py -q -x --timeout=600 > $O/pytest_convnet.log 2>&1; echo "pytest rc=$?" >> $O/pytest_convnet.log; tail -15 $O/pytest_convnet.log
timeout 600 python bench.py --workload convnet --steps 10 --warmup 2 > $O/bench_convnet.json 2> $O/bench_convnet.err; echo "bench convnet rc=$?"; cut -c1-3000 $O/bench_convnet.json; tail -3 $O/bench_convnet.err
timeout 400 python tools/bench_model.py --single-rank-group --steps 5 --warmup 2 > $O/model_eager_srg.json 2> $O/model_eager_srg.err; echo "model eager srg rc=$?"; cat $O/model_eager_srg.json; tail -3 $O/model_eager_srg.err
timeout 400 python tools/bench_model.py --single-rank-group --graph --steps 10 --warmup 3 > $O/model_graph_srg.json 2> $O/model_graph_srg.err; echo "model graph srg rc=$?"; cat $O/model_graph_srg.json; tail -3 $O/model_graph_srg.err
timeout 400 python tools/bench_model.py --graph --steps 10 --warmup 3 > $O/model_graph.json 2> $O/model_graph.err; echo "model graph rc=$?"; cat $O/model_graph.json; tail -3 $O/model_graph.err
timeout 600 python tools/bench_stages.py > $O/stage_shapes.json 2> $O/stage_shapes.err; echo "stages rc=$?"; tail -3 $O/stage_shapes.err
