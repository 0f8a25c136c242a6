// Test-loop accumulation of per-crop predictions into per-point sums (the tail of the GPU dataloader stage).
//   cumulate_probs   /root/reference/pytorch/tool/test.py:330-352
//       cum_dict['probs'][inds, ...] += pred                                   (smooth None)
//       cum_dict['probs'][inds, ...] = smooth * cum_dict['probs'][inds, ...] + (1 - smooth) * pred
//       cum_dict['probs_last'][inds, ...] = pred
// `inds` is the concatenation of the crops of one batch (:225-229) and crops overlap, so it holds duplicates; an indexed `+=` is a gather,
// an add and an indexed ASSIGNMENT: for a duplicated point ONE row of pred is added (on the CPU the last one, on CUDA whichever store
// lands last).  Here the rule is the CPU's, deterministically: row r counts iff no later row carries the same index.
//   pass 1: last[inds[r]] = max(r)     (int atomics on a scratch array the size of the cloud, -1 initialised)
//   pass 2: winners update their point's row, lane = class
#include "cbl_common.h"

namespace {

__global__ __launch_bounds__(256) void cum_last_kernel(int m, int n, const long long* __restrict__ inds, int* __restrict__ last)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    const long long j = inds[r];
    if (j >= 0 && j < n) atomicMax(last + j, r);
}

// mode 0: probs += pred; 1: probs = smooth * probs + (1 - smooth) * pred; 2: probs = pred
__global__ __launch_bounds__(256) void cum_apply_kernel(int m, int n, int ncls, const long long* __restrict__ inds, const int* __restrict__ last,
                                                        const float* __restrict__ pred, float smooth, int mode, float* __restrict__ probs)
{
    const long long total = (long long)m * ncls;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int r = (int)(e / ncls), c = (int)(e - (long long)r * ncls);
        const long long j = inds[r];
        if (j < 0 || j >= n || last[j] != r) continue;
        float* dst = probs + (size_t)j * ncls + c;
        const float v = pred[e];
        *dst = mode == 0 ? *dst + v : mode == 1 ? smooth * *dst + (1.0f - smooth) * v : v;
    }
}

}  // namespace

CBL_EXPORT int cbl_cumulate_probs(int n, int num_classes, int m, const long long* inds, const float* pred, float smooth, int mode, float* probs,
                                  int* scratch_n, void* stream)
{
    if (n < 0 || m < 0 || num_classes <= 0 || mode < 0 || mode > 2) return CBL_ERR_BAD_ARG;
    if (m == 0 || n == 0) return CBL_OK;
    if (!inds || !pred || !probs || !scratch_n) return CBL_ERR_BAD_ARG;
    hipStream_t st = cbl_stream(stream);
    if (hipMemsetAsync(scratch_n, 0xff, sizeof(int) * (size_t)n, st) != hipSuccess) return cbl_status();
    hipLaunchKernelGGL(cum_last_kernel, dim3(cbl_div_up(m, 256)), dim3(256), 0, st, m, n, inds, scratch_n);
    hipLaunchKernelGGL(cum_apply_kernel, dim3(cbl_grid_for((long long)m * num_classes, 256)), dim3(256), 0, st, m, n, num_classes, inds, scratch_n, pred, smooth, mode, probs);
    return cbl_status();
}
