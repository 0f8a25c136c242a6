// TEST INFRASTRUCTURE: the product's K4 gather kernel (contrastboundary_amd/csrc/k4_rows_pipe.h) compiled for the host against the fibre emulation of
// tests/host_emul/wave.  Built and called by tests/test_k4_rows_host.py.
#include <hip/hip_runtime.h>

#include "cbl_common.h"

namespace {
#include "k4_rows_pipe.h"
}

extern "C" int k4_rows(unsigned grid, unsigned n, int c, int stride, int off, const float* go, const int* order, const int* inv_start, const int* inv_src,
                       float* gi)
{
    hipLaunchKernelGGL(grouping_bwd_csr_rows_kernel, dim3(grid), dim3(K4_ROWS_BLOCK), 0, nullptr, n, c, stride, off, go, order, inv_start, inv_src, gi);
    return 0;
}
