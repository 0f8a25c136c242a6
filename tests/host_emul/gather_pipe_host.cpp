// TEST INFRASTRUCTURE: the product's pipelined gather kernel (contrastboundary_amd/csrc/query_group_pipe.h) compiled for the host against the fibre
// emulation of tests/host_emul/wave, with one C entry per shape.  Built and called by tests/test_gather_pipe_host.py.
#include <hip/hip_runtime.h>

#include "cbl_common.h"

namespace {
#include "query_group_pipe.h"
}

#define ENTRY(NAME, C4T, PR)                                                                                                                          \
    extern "C" int NAME(unsigned grid, unsigned npieces, const float* xyz, const float* new_xyz, const float* feat, const int* idx, const int* order, \
                        float* out)                                                                                                                   \
    {                                                                                                                                                 \
        hipLaunchKernelGGL((query_group_lds_pipe<C4T, PR>), dim3(grid), dim3(256), 0, nullptr, npieces, xyz, new_xyz,                                 \
                           reinterpret_cast<const float4*>(feat), idx, order, out);                                                                   \
        return 0;                                                                                                                                     \
    }
ENTRY(gather_pipe_c32_k8, 8, 8)
ENTRY(gather_pipe_c32_k16, 8, 16)
ENTRY(gather_pipe_c64_k8, 16, 8)
ENTRY(gather_pipe_c64_k16, 16, 16)
