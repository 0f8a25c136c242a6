#!/bin/bash
# GPU side of the interpreter-free checks (a few seconds in all): run from the repository root on the GPU box
#   /usr/local/graft/bin/gpurun --timeout 120 -- 'bash tools/device_check/run.sh'
# the binaries are built HERE beforehand (tools/device_check/build.sh: hipcc cross-compiles); compare afterwards with compare.py / float_compare.py / the fnv lines
mkdir -p gpurun_out
for p in cbl_check path_check float_check layer_check wide_check; do
    timeout 60 ./tools/device_check/${p}_dev gpurun_out/${p}_dev.bin > gpurun_out/${p}_dev.log 2>&1
    echo "exit $?" >> gpurun_out/${p}_dev.log
    tail -2 gpurun_out/${p}_dev.log
done
rm -f gpurun_out/path_check_dev.bin                                   # 10 MB of raw tables: the log's fnv lines are what is compared
