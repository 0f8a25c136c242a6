#!/bin/bash
# Builds the interpreter-free programs of this directory from the repository root:
#   *_dev   hipcc, against contrastboundary_amd/lib/libcbl_amd.so (run them on the GPU box: `gpurun -- ./tools/device_check/<name>_dev ...`, seconds per call)
#   *_host  g++ -DHOST_EMULATED, against the host-emulated build of the same kernel sources (tests/host_emul/full_library.py) — cbl_check and path_check only
# then:  python tools/device_check/compare.py gpurun_out/cbl_check_dev.bin /tmp/cbl_check_host.bin      (contrast kernels: tolerances)
#        diff <(grep fnv gpurun_out/path_check_dev.log) <(grep fnv /tmp/path_check_host.log)            (bit-exact kernels: byte-identical)
set -e
cd "$(dirname "$0")/../.."
python -c "from contrastboundary_amd import build; build.build()"
HOSTLIB=$(python -c "from tests.host_emul import full_library as f; import os; print(os.path.dirname(f.build()))")
D=tools/device_check
for p in cbl_check path_check float_check layer_check wide_check gather_time; do
    hipcc --offload-arch=gfx950 -O2 -std=c++17 $D/$p.cpp -o $D/${p}_dev -Lcontrastboundary_amd/lib -lcbl_amd -Wl,-rpath,'$ORIGIN/../../contrastboundary_amd/lib'
done
hipcc --offload-arch=gfx950 -O2 -std=c++17 $D/cbl_time.cpp -o $D/cbl_time_dev -ldl
for p in cbl_check path_check float_check layer_check wide_check; do
    g++ -std=c++17 -O1 -DHOST_EMULATED $D/$p.cpp -o /tmp/${p}_host -L"$HOSTLIB" -lcbl_amd_host -Wl,-rpath,"$HOSTLIB"
done
echo "built: $D/*_dev, /tmp/cbl_check_host, /tmp/path_check_host"
