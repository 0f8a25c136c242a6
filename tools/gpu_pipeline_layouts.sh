#!/bin/bash
# A/B of hotpath.Pipeline layouts in ONE gpurun call (box-to-box spread is larger than the differences).  Output: gpurun_out/layouts/*.json
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/layouts
run() {  # tag, env...
    tag=$1; shift
    for rep in 1 2; do
        env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra --no-legs --no-gather-200k --steps 200 --warmup 10 > gpurun_out/layouts/${tag}_$rep.json 2> gpurun_out/layouts/${tag}_$rep.err
        python - "$tag" "$rep" <<'P'
import json, sys
tag, rep = sys.argv[1:3]
try:
    d = json.loads(open("gpurun_out/layouts/%s_%s.json" % (tag, rep)).read().strip().splitlines()[-1])
    print(tag, rep, "ms_per_step %.4f" % d["ms_per_step"])
except Exception as e:
    print(tag, rep, "FAILED", e)
P
    done
}
run tables2 CBL_PIPELINE_LAYOUT=tables CBL_PIPELINE_SLOTS=2
run split2 CBL_PIPELINE_LAYOUT=split CBL_PIPELINE_SLOTS=2
run split3 CBL_PIPELINE_LAYOUT=split CBL_PIPELINE_SLOTS=3
run early2 CBL_PIPELINE_LAYOUT=split_early CBL_PIPELINE_SLOTS=2
run early3 CBL_PIPELINE_LAYOUT=split_early CBL_PIPELINE_SLOTS=3
run tables3 CBL_PIPELINE_LAYOUT=tables CBL_PIPELINE_SLOTS=3
run split3q8 CBL_PIPELINE_LAYOUT=split CBL_PIPELINE_SLOTS=3 GPU_MAX_HW_QUEUES=8
run tables2b CBL_PIPELINE_LAYOUT=tables CBL_PIPELINE_SLOTS=2
