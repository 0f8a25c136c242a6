#!/bin/bash
# round 6 measurement set: every number DESIGN.md / README.md quote for this round comes from one run of this script (outputs: gpurun_out/r06final/)
# usage: bash tools/gpu_r06_final.sh <commit>
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06final
mkdir -p $O
cd $GRAFT_REPO_ROOT
COMMIT=${1:-unknown}
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/device.txt 2>&1
# ---- HBM-side traffic of the headline's kernels and of the 200 000-point gather: separate --pmc passes, --kernel-trace only
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-pipeline > $O/pmc_$c.log 2>&1; echo "pmc $c rc=$?")
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmcg_$c -o pmc -- python $GRAFT_REPO_ROOT/tools/gather_200k.py > $O/pmcg_$c.log 2>&1; echo "pmc gather_200k $c rc=$?")
done
F=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_summary.py "$F" "$W" $O/pmc_traffic.json > $O/pmc_summary.txt 2>&1
F=$(find $O/pmcg_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $O/pmcg_WRITE_SIZE -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_summary.py "$F" "$W" $O/pmc_traffic_gather200k.json > $O/pmc_summary_gather200k.txt 2>&1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmcg_FETCH_SIZE $O/pmcg_WRITE_SIZE
[ -f $O/pmc_traffic.json ] && python - "$O" "$COMMIT" <<'PY'
import json, os, sys
O, commit = sys.argv[1], sys.argv[2]
d = json.load(open(O + "/pmc_traffic.json"))
d.setdefault("_meta", {})["commit"] = commit; d["_meta"]["round"] = 6
d["_meta"]["kernels"] = "the kernel set of commit %s (same run of tools/gpu_r06_final.sh as the bench lines)" % commit
g = O + "/pmc_traffic_gather200k.json"
if os.path.exists(g):
    gd = json.load(open(g))
    k = [x for x in gd if x.startswith("query_group_lds")]
    if k:
        e = dict(gd[k[0]]); e["kernel"] = k[0]
        e["note"] = "python tools/gather_200k.py under the same two --pmc passes: every launch of the kernel in that process is the N = 200 000 gather (926 MB algorithmic, past the Infinity Cache)"
        d["gather_200k"] = e
json.dump(d, open(O + "/pmc_traffic.json", "w"), indent=1); json.dump(d, open("profiles/r06_pmc_traffic.json", "w"), indent=1)
PY
# ---- instruction mix of the headline step's kernels (the search's vector-issue fraction: bench.py roofline.search) and of the attention layer's passes
bash tools/gpu_pmc_mix.sh $COMMIT r06_headline > $O/pmc_mix_headline.txt 2>&1; cp gpurun_out/pmc_r06_headline/pmc_mix.json profiles/r06_pmc_instruction_mix.json 2>/dev/null
bash tools/gpu_pmc_any.sh r06_pt_layer "^pt_|^triple" python $GRAFT_REPO_ROOT/tools/pt_layer_time.py 40960 16 64 > $O/pmc_pt_layer.txt 2>&1; cp gpurun_out/pmc_r06_pt_layer/pmc_mix.json $O/pmc_pt_layer.json 2>/dev/null
bash tools/gpu_pmc_any.sh r06_kpconv "kpconv|query_group|grouping_bwd|contrast" python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-pipeline > $O/pmc_kpconv.txt 2>&1; cp gpurun_out/pmc_r06_kpconv/pmc_mix.json $O/pmc_kpconv.json 2>/dev/null
# ---- the three bench lines (the default one carries the pt_block and convnet legs) and their kernel statistics
timeout 900 python bench.py --steps 50 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 400 python bench.py --block pt --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_pt.json 2> $O/bench_pt.err; echo "bench pt rc=$?"
timeout 600 python bench.py --workload convnet --steps 40 --warmup 5 > $O/bench_convnet.json 2> $O/bench_convnet.err; echo "bench convnet rc=$?"
prof() { tag=$1; shift; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o $tag -- "$@" > $O/prof_$tag.log 2>&1; echo "rocprof $tag rc=$?"); f=$(find $O/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${tag}_kernel_stats.csv; rm -rf $O/prof_$tag; }
prof bench python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-legs --no-gather-200k
prof bench_pt python $GRAFT_REPO_ROOT/bench.py --block pt --steps 30 --warmup 5 --no-cpu-baseline --no-extra
prof bench_convnet python $GRAFT_REPO_ROOT/bench.py --workload convnet --steps 20 --warmup 3 --no-cpu-baseline --no-extra
prof pt_layer python $GRAFT_REPO_ROOT/tools/pt_layer_time.py 40960 16 64
prof wide_layer python $GRAFT_REPO_ROOT/tools/pt_layer_time.py 640 16 256
prof pt_layer_k8c32 python $GRAFT_REPO_ROOT/tools/pt_layer_time.py 40960 8 32
# ---- the layer alone (both full-resolution shapes and the three wide ones), the radius search alone, the sampler, the full network
for s in "40960 16 64" "40960 8 32" "2560 16 128" "640 16 256" "160 16 512"; do timeout 300 python tools/pt_layer_time.py $s --graph 2>/dev/null | tail -1; done > $O/pt_layer_time.jsonl
timeout 300 python tools/radius_time.py > $O/radius_time.json 2>/dev/null
timeout 120 python tools/fps_time.py > $O/fps_time.json 2>/dev/null
model() { tag=$1; shift; timeout 300 python tools/bench_model.py "$@" 2> $O/model_$tag.err | grep '^{' | tail -1 > $O/model_$tag.json; echo "model $tag rc=$?"; }
model graph --graph --steps 10 --warmup 3
model graph_d1 --graph --depth 1 --steps 10 --warmup 3
model graph_4scenes --graph --scenes 4 --steps 6 --warmup 2
model graph_8scenes --graph --scenes 8 --steps 4 --warmup 2
model srg_flat --single-rank-group --graph --steps 10 --warmup 3
model srg_flat_4scenes --single-rank-group --graph --scenes 4 --steps 6 --warmup 2
model srg_hook --single-rank-group --graph --hook-reducer --steps 10 --warmup 3
CBL_GEO_STREAMS=unprobed_first model srg_flat_unprobed_streams --single-rank-group --graph --steps 10 --warmup 3
model eager_srg_flat --single-rank-group --steps 10 --warmup 3
cat $O/model_graph.json $O/model_graph_d1.json $O/model_graph_4scenes.json $O/model_graph_8scenes.json $O/model_srg_flat.json $O/model_srg_flat_4scenes.json $O/model_srg_hook.json $O/model_srg_flat_unprobed_streams.json $O/model_eager_srg_flat.json > $O/bench_model.jsonl
prof model python $GRAFT_REPO_ROOT/tools/bench_model.py --graph --steps 10 --warmup 3
timeout 300 python bench.py --workload stage_shapes > $O/bench_stage_shapes.json 2> $O/bench_stage_shapes.err; echo "bench stage_shapes rc=$?"
timeout 600 python tools/bench_stages.py > $O/stage_shapes.json 2> $O/stage_shapes.err; echo "stages rc=$?"
bash tools/gpu_pmc_any.sh r06_radius "radius|grid_" python $GRAFT_REPO_ROOT/tools/radius_time.py > $O/pmc_radius.txt 2>&1; cp gpurun_out/pmc_r06_radius/pmc_mix.json $O/pmc_radius.json 2>/dev/null
timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu > $O/fuzz_device.log 2>&1; echo "fuzz rc=$?"
ls -la $O | head -80
