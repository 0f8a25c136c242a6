#!/bin/bash
# GPU side of the Python-free check (seconds: no interpreter, no torch import): run from the repository root on the GPU box
#   /usr/local/graft/bin/gpurun --timeout 120 -- 'bash tools/device_check/run.sh'
# the binary is built HERE beforehand (hipcc cross-compiles): see the header of cbl_check.cpp
mkdir -p gpurun_out
timeout 90 ./tools/device_check/cbl_check_dev gpurun_out/cbl_check_dev.bin > gpurun_out/cbl_check_dev.log 2>&1
echo "exit $?" >> gpurun_out/cbl_check_dev.log
tail -3 gpurun_out/cbl_check_dev.log
