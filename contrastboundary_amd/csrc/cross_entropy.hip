// The criterion's first term: nn.CrossEntropyLoss(ignore_index) over the per-point logits  /root/reference/pytorch/model/pointtransformer_seg.py:20-22
//   loss = mean over the points whose label is not ignore_index of  logsumexp(logits[i, :]) - logits[i, target[i]]
// The library computes it as log_softmax + nll_loss; its nll_loss reduction is ONE workgroup walking every point, forward and backward (306 us + 277 us on
// 327 680 points x 13 classes, 5 us of traffic).  Here: one lane per point (k <= 64 classes: a row is a few registers), per-workgroup partial sums of the
// loss and of the valid count, a finalize launch that sums them in fp64 in a fixed order (deterministic, no atomics), and a backward pass that recomputes the
// softmax:  grad_logits[i, c] = (softmax[i, c] - [c == target[i]]) * grad_loss / count  (0 for ignored points).
#include "cbl_common.h"
#include "../../include/cbl_amd.h"

namespace {

constexpr int XE_BLOCK = 256, XE_MAX_BLOCKS = 1024, XE_MAX_K = 64;

// row maximum and log of the sum of exponentials, in the order of the classes (expf / logf as the library's log_softmax: within 1 ulp-level of it)
__device__ __forceinline__ void xe_row(const float* __restrict__ z, int k, float& mx, float& lse)
{
    mx = z[0];
    for (int c = 1; c < k; c++) mx = fmaxf(mx, z[c]);
    float s = 0.f;
    for (int c = 0; c < k; c++) s += expf(z[c] - mx);
    lse = mx + logf(s);
}

__global__ __launch_bounds__(XE_BLOCK) void xe_forward_kernel(long long n, int k, const float* __restrict__ logits, const long long* __restrict__ target,
                                                              long long ignore_index, double* __restrict__ partial)
{
    __shared__ double red[2][XE_BLOCK / 64];
    double sum = 0.0, cnt = 0.0;
    for (long long i = (long long)blockIdx.x * XE_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * XE_BLOCK) {
        const long long t = target[i];
        if (t == ignore_index) continue;
        if (t < 0 || t >= k) { sum += (double)NAN; continue; }       // a label outside [0, k) that is not ignore_index (a mapping bug: 255 against another ignore label):
                                                                     // nn.CrossEntropyLoss device-asserts; here the loss turns NaN — loud, and legal inside a replayed hipGraph
        float mx, lse;
        xe_row(logits + i * k, k, mx, lse);
        sum += (double)(lse - logits[i * k + t]);
        cnt += 1.0;
    }
    for (int s = 32; s >= 1; s >>= 1) { sum += __shfl_xor(sum, s); cnt += __shfl_xor(cnt, s); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = sum; red[1][wave] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < XE_BLOCK / 64; w++) { a += red[0][w]; b += red[1][w]; }
        partial[2 * blockIdx.x] = a; partial[2 * blockIdx.x + 1] = b;
    }
}

// loss = sum / count (nan for count = 0, like the library's mean over nothing); stats = {sum, count} for the backward pass
__global__ __launch_bounds__(64) void xe_finalize_kernel(int nblocks, const double* __restrict__ partial, float* __restrict__ loss, float* __restrict__ stats)
{
    double a = 0.0, b = 0.0;
    for (int j = threadIdx.x; j < nblocks; j += 64) { a += partial[2 * j]; b += partial[2 * j + 1]; }
    for (int s = 32; s >= 1; s >>= 1) { a += __shfl_xor(a, s); b += __shfl_xor(b, s); }
    if (threadIdx.x == 0) { loss[0] = (float)(a / b); stats[0] = (float)a; stats[1] = (float)b; }
}

__global__ __launch_bounds__(XE_BLOCK) void xe_backward_kernel(long long n, int k, const float* __restrict__ logits, const long long* __restrict__ target,
                                                               long long ignore_index, const float* __restrict__ stats, const float* __restrict__ grad_loss,
                                                               float* __restrict__ grad_logits)
{
    const float scale = grad_loss[0] / stats[1];
    for (long long i = (long long)blockIdx.x * XE_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * XE_BLOCK) {
        const long long t = target[i];
        float* g = grad_logits + i * k;
        if (t == ignore_index || t < 0 || t >= k) { for (int c = 0; c < k; c++) g[c] = 0.f; continue; }
        float mx, lse;
        const float* z = logits + i * k;
        xe_row(z, k, mx, lse);
        for (int c = 0; c < k; c++) g[c] = (expf(z[c] - lse) - (c == (int)t ? 1.f : 0.f)) * scale;
    }
}

}  // namespace

CBL_EXPORT size_t cbl_cross_entropy_workspace_bytes(long long n) { (void)n; return sizeof(double) * 2 * XE_MAX_BLOCKS + 256; }

CBL_EXPORT int cbl_cross_entropy_forward(long long n, int k, const float* logits, const long long* target, long long ignore_index, float* loss, float* stats,
                                         void* workspace, size_t workspace_bytes, void* stream)
{
    if (n < 0 || k < 1) return CBL_ERR_BAD_ARG;
    if (k > XE_MAX_K) return CBL_ERR_UNSUPPORTED;
    if (!loss || !stats || !workspace || (n > 0 && (!logits || !target))) return CBL_ERR_BAD_ARG;
    if (workspace_bytes < cbl_cross_entropy_workspace_bytes(n)) return CBL_ERR_WORKSPACE;
    hipStream_t st = cbl_stream(stream);
    double* partial = reinterpret_cast<double*>(workspace);
    const int nblocks = (int)cbl_grid_for(n > 0 ? n : 1, XE_BLOCK, XE_MAX_BLOCKS);
    hipLaunchKernelGGL(xe_forward_kernel, dim3(nblocks), dim3(XE_BLOCK), 0, st, n, k, logits, target, ignore_index, partial);
    hipLaunchKernelGGL(xe_finalize_kernel, dim3(1), dim3(64), 0, st, nblocks, partial, loss, stats);
    return cbl_status();
}

CBL_EXPORT int cbl_cross_entropy_backward(long long n, int k, const float* logits, const long long* target, long long ignore_index, const float* stats,
                                          const float* grad_loss, float* grad_logits, void* stream)
{
    if (n < 0 || k < 1) return CBL_ERR_BAD_ARG;
    if (k > XE_MAX_K) return CBL_ERR_UNSUPPORTED;
    if (n == 0) return CBL_OK;
    if (!logits || !target || !stats || !grad_loss || !grad_logits) return CBL_ERR_BAD_ARG;
    hipLaunchKernelGGL(xe_backward_kernel, dim3(cbl_grid_for(n, XE_BLOCK, 4096)), dim3(XE_BLOCK), 0, cbl_stream(stream), n, k, logits, target, ignore_index, stats,
                       grad_loss, grad_logits);
    return cbl_status();
}
