#!/bin/bash
# copy the judged summaries of gpurun_out/r04final (one run of tools/gpu_r04_final.sh) into profiles/ under round-4 names
set -u
S=gpurun_out/r04final; D=profiles
cp $S/bench.json $D/r04_bench.json; cp $S/bench_pt.json $D/r04_bench_pt.json; cp $S/bench_convnet.json $D/r04_bench_convnet.json
cp $S/bench_kernel_stats.csv $D/r04_bench_kernel_stats.csv; cp $S/bench_pt_kernel_stats.csv $D/r04_bench_pt_kernel_stats.csv
cp $S/bench_convnet_kernel_stats.csv $D/r04_bench_convnet_kernel_stats.csv; cp $S/pt_layer_kernel_stats.csv $D/r04_pt_layer_kernel_stats.csv
cp $S/model_kernel_stats.csv $D/r04_model_kernel_stats.csv
cp $S/pmc_traffic.json $D/r04_pmc_traffic.json; cp $S/pmc_pt_layer.json $D/r04_pmc_pt_layer.json; cp $S/pmc_radius.json $D/r04_pmc_radius.json
cp $S/stage_shapes.json $D/r04_stage_shapes.json; cp $S/fps_time.json $D/r04_fps_time.json; cp $S/radius_time.json $D/r04_radius_time.json
python3 - <<'PY'
import json
S, D = "gpurun_out/r04final/", "profiles/"
rows = []
for f, tag in (("pt_layer_40960_16_64.json", None), ("pt_layer_40960_8_32.json", None)):
    rows.append(json.loads(open(S + f).read().strip().splitlines()[-1]))
json.dump(rows, open(D + "r04_pt_layer_time.json", "w"), indent=1)
with open(D + "r04_bench_model.jsonl", "w") as out:
    for f, tag in (("model_graph.json", "graph, depth 2"), ("model_graph_d1.json", "graph, depth 1"), ("model_graph_4scenes.json", "graph, 4 scenes"), ("model_graph_srg.json", "graph, one-rank RCCL group + gradient reducer")):
        for line in open(S + f).read().splitlines():
            if line.startswith("{"):
                d = json.loads(line); d["run"] = tag; out.write(json.dumps(d) + "\n")
PY
rm -f $D/r04_bench_pt_mid.json $D/r04_pt_layer_kernel_stats_mid.csv $D/r04_pmc_pt_layer_mid.json
ls $D | grep r04
