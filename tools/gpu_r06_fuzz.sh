#!/bin/bash
# tests/test_gpu_fuzz.py on the device with the shipped library (must pass) and with a build whose contrast kernels are those of 301ff19^ (CBL_AMD_LIB;
# the two regression inputs must FAIL there): gpurun_out/r06fuzz/
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r06fuzz; mkdir -p $O
python -m pytest tests/test_gpu_fuzz.py -q -m gpu > $O/shipped.log 2>&1; echo "shipped rc=$?" | tee -a $O/shipped.log
if [ -f contrastboundary_amd/lib/libcbl_amd_prefix.so ]; then
  CBL_AMD_LIB=$PWD/contrastboundary_amd/lib/libcbl_amd_prefix.so python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "finite or keep_their_sum or cbl" > $O/prefix.log 2>&1; echo "prefix rc=$? (non-zero expected)" | tee -a $O/prefix.log
fi
tail -5 $O/shipped.log; grep -E "^(FAILED|PASSED|ERROR)|passed|failed" $O/prefix.log | tail -12
