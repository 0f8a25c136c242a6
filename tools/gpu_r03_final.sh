#!/bin/bash
# round 3 measurement set: every number DESIGN.md / README.md quote comes from one run of this script (outputs: gpurun_out/r03final/)
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03final
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/device.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-pipeline > $O/pmc_$c.log 2>&1; echo "pmc $c rc=$?")
done
F=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_summary.py "$F" "$W" $O/pmc_traffic.json > $O/pmc_summary.txt 2>&1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
# the bench lines below read their traffic figures from this run's counters, tagged with the commit given as $1
[ -f $O/pmc_traffic.json ] && python - "$O/pmc_traffic.json" "${1:-unknown}" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); d.setdefault("_meta", {})["commit"] = sys.argv[2]; d["_meta"]["round"] = 3
d["_meta"]["kernels"] = "the kernel set of commit %s (same run of tools/gpu_r03_final.sh as the bench lines)" % sys.argv[2]
json.dump(d, open(sys.argv[1], "w"), indent=1); json.dump(d, open("profiles/r03_pmc_traffic.json", "w"), indent=1)
PY
timeout 600 python bench.py --steps 50 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 400 python bench.py --block pt --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_pt.json 2> $O/bench_pt.err; echo "bench pt rc=$?"
timeout 600 python bench.py --workload convnet --steps 20 --warmup 3 > $O/bench_convnet.json 2> $O/bench_convnet.err; echo "bench convnet rc=$?"
prof() { tag=$1; shift; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o $tag -- python $GRAFT_REPO_ROOT/bench.py "$@" > $O/prof_$tag.log 2>&1; echo "rocprof $tag rc=$?"); f=$(find $O/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${tag}_kernel_stats.csv; rm -rf $O/prof_$tag; }
prof bench --steps 50 --warmup 5 --no-cpu-baseline
prof bench_pt --block pt --steps 30 --warmup 5 --no-cpu-baseline --no-extra
prof bench_convnet --workload convnet --steps 20 --warmup 3 --no-cpu-baseline --no-extra
timeout 300 python tools/bench_model.py --graph --steps 10 --warmup 3 > $O/model_graph.json 2> $O/model_graph.err; echo "model graph rc=$?"
timeout 300 python tools/bench_model.py --graph --depth 1 --steps 10 --warmup 3 > $O/model_graph_d1.json 2> $O/model_graph_d1.err; echo "model graph depth1 rc=$?"
timeout 300 python tools/bench_model.py --graph --scenes 4 --steps 6 --warmup 2 > $O/model_graph_4scenes.json 2> $O/model_graph_4.err; echo "model graph 4 scenes rc=$?"
timeout 300 python tools/bench_model.py --single-rank-group --prefetch --steps 6 --warmup 2 > $O/model_eager_srg.json 2> $O/model_eager_srg.err; echo "model eager one-rank RCCL rc=$?"
timeout 300 python tools/bench_model.py --single-rank-group --graph --steps 10 --warmup 3 > $O/model_graph_srg.json 2> $O/model_graph_srg.err; echo "model graph one-rank RCCL rc=$?"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_model -o model -- python $GRAFT_REPO_ROOT/tools/bench_model.py --graph --steps 10 --warmup 3 > $O/prof_model.log 2>&1; echo "rocprof model rc=$?"); f=$(find $O/prof_model -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/model_kernel_stats.csv; rm -rf $O/prof_model
timeout 600 python tools/bench_stages.py > $O/stage_shapes.json 2> $O/stage_shapes.err; echo "stages rc=$?"
timeout 120 python tools/fps_time.py > $O/fps_time.json 2>/dev/null
ls -la $O | head -40
