// a4 / a5, dense part: train-mode BatchNorm1d (+ ReLU) over (rows, C) activations with rows = n or n*K — the layers between the
// neighbourhood kernels of the blocks (/root/reference/pytorch/model/blocks.py:25-28,38-40,70,74,126-134: BatchNorm1d after every
// Linear, most of them followed by ReLU).  The library path runs them as statistics + transform + clamp kernels forward and
// threshold + reduce + element kernels backward at ~1.4 TB/s; here the same mathematics is 2 streaming passes forward and 2
// backward with the ReLU folded in:
//   forward   pass 1: per-workgroup partial sums of x and x^2 per channel (fp32 over <= ~100 rows per lane, combined in fp64)
//             finalize (16 channels per workgroup): mean, biased variance -> invstd; running statistics updated like nn.BatchNorm1d
//             (momentum, unbiased variance)
//             pass 2: y = max(0, (x - mean) * invstd * weight + bias)
//   backward  pass 1: partial sums of g and g * xhat per channel, g = grad_y masked by (y > 0) recomputed from x
//             finalize: grad_weight = sum g*xhat, grad_bias = sum g
//             pass 2: grad_x = weight * invstd * (g - mean(g) - xhat * mean(g*xhat))
// Row-major (rows, C): a lane owns VEC consecutive channels and walks rows, so every access is a coalesced row segment.
#include "cbl_common.h"
#include "wave_ops.h"

namespace {

constexpr int BN_BLOCK = 256;
constexpr int BN_MAX_BLOCKS = 512;

struct BnShape { int vec, tpr, slots, nblocks; long long rows_per_block; };
inline BnShape bn_shape(long long rows, int C)
{
    BnShape s;
    s.vec = (C % 4 == 0) ? 4 : 1;
    s.tpr = C / s.vec;                                               // lanes per row
    s.slots = BN_BLOCK / s.tpr;                                      // rows in flight per workgroup
    long long nb = (rows + 63) / 64;
    s.nblocks = (int)(nb < 1 ? 1 : (nb > BN_MAX_BLOCKS ? BN_MAX_BLOCKS : nb));
    s.rows_per_block = (rows + s.nblocks - 1) / s.nblocks;
    return s;
}

// partial[b][0][c] = sum over the workgroup's rows of A(r,c), partial[b][1][c] = sum of B(r,c)
//   MODE 0 (forward):  A = x, B = x*x
//   MODE 1 (backward): A = g, B = g * xhat   with g = gy * (relu ? y > 0 : 1), xhat = (x - mean) * invstd, y = xhat * w + b (+ residual)
template <int VEC, int MODE>
__global__ __launch_bounds__(BN_BLOCK) void bn_partial_kernel(long long rows, int C, int tpr, int slots, long long rows_per_block,
                                                              const float* __restrict__ x, const float* __restrict__ gy,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              const float* __restrict__ weight, const float* __restrict__ bias, int relu,
                                                              const float* __restrict__ residual, float* __restrict__ partial)
{
    __shared__ float red[2][BN_BLOCK][VEC];
    const int tid = threadIdx.x;
    const int cq = tid % tpr, slot = tid / tpr;
    const bool live = slot < slots;
    const int c0 = cq * VEC;
    float m[VEC], is[VEC], w[VEC], b[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) {
        m[v] = (MODE == 1) ? mean[c0 + v] : 0.f; is[v] = (MODE == 1) ? invstd[c0 + v] : 0.f;
        w[v] = (MODE == 1 && weight) ? weight[c0 + v] : 1.f; b[v] = (MODE == 1 && bias) ? bias[c0 + v] : 0.f;
    }
    float a0[VEC], a1[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) a0[v] = a1[v] = 0.f;
    const long long r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    if (live) {
        for (long long r = r0 + slot; r < r1; r += slots) {
            float xv[VEC], gv[VEC], rv[VEC];
#pragma unroll
            for (int v = 0; v < VEC; v++) rv[v] = 0.f;
            if (VEC == 4) {
                const float4 t = *reinterpret_cast<const float4*>(x + r * C + c0);
                xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
                if (MODE == 1) { const float4 u = *reinterpret_cast<const float4*>(gy + r * C + c0); gv[0] = u.x; gv[1] = u.y; gv[2] = u.z; gv[3] = u.w; }
                if (MODE == 1 && residual) { const float4 u = *reinterpret_cast<const float4*>(residual + r * C + c0); rv[0] = u.x; rv[1] = u.y; rv[2] = u.z; rv[3] = u.w; }
            } else {
                xv[0] = x[r * C + c0];
                if (MODE == 1) gv[0] = gy[r * C + c0];
                if (MODE == 1 && residual) rv[0] = residual[r * C + c0];
            }
#pragma unroll
            for (int v = 0; v < VEC; v++) {
                if (MODE == 0) { a0[v] += xv[v]; a1[v] += xv[v] * xv[v]; }
                else {
                    const float xh = (xv[v] - m[v]) * is[v];
                    const float g = (relu && !((xh * w[v] + b[v]) + rv[v] > 0.f)) ? 0.f : gv[v];
                    a0[v] += g; a1[v] += g * xh;
                }
            }
        }
    }
#pragma unroll
    for (int v = 0; v < VEC; v++) { red[0][tid][v] = a0[v]; red[1][tid][v] = a1[v]; }
    __syncthreads();
    if (live && slot == 0) {
#pragma unroll
        for (int v = 0; v < VEC; v++) {
            double s0 = 0.0, s1 = 0.0;
            for (int s = 0; s < slots; s++) { s0 += (double)red[0][s * tpr + cq][v]; s1 += (double)red[1][s * tpr + cq][v]; }
            partial[((size_t)blockIdx.x * 2 + 0) * C + c0 + v] = (float)s0;
            partial[((size_t)blockIdx.x * 2 + 1) * C + c0 + v] = (float)s1;
        }
    }
}

// sum of the per-workgroup partials of 16 channels by one 256-lane workgroup: lane (channel cs, slice js) adds every 16th partial,
// the 16 slices are combined through LDS in fp64.  (One lane per channel walking all partials was a 70 us dependent-load chain.)
__device__ __forceinline__ void bn_sum_partials(int C, int nblocks, const float* __restrict__ partial, int c, int js, double (*red)[16][2],
                                                double& s0, double& s1)
{
    double a0 = 0.0, a1 = 0.0;
    if (c < C) {
#pragma unroll 8
        for (int b = js; b < nblocks; b += 16) { a0 += (double)partial[((size_t)b * 2) * C + c]; a1 += (double)partial[((size_t)b * 2 + 1) * C + c]; }
    }
    red[js][threadIdx.x & 15][0] = a0; red[js][threadIdx.x & 15][1] = a1;
    __syncthreads();
    s0 = s1 = 0.0;
    if (js == 0)
        for (int j = 0; j < 16; j++) { s0 += red[j][threadIdx.x & 15][0]; s1 += red[j][threadIdx.x & 15][1]; }
}

// forward finalize: batch statistics, running statistics (nn.BatchNorm1d: momentum, unbiased variance)
__global__ __launch_bounds__(256) void bn_fwd_finalize_kernel(long long rows, int C, int nblocks, const float* __restrict__ partial, float eps,
                                                              float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                              long long* __restrict__ num_batches_tracked,
                                                              float* __restrict__ mean, float* __restrict__ invstd)
{
    __shared__ double red[16][16][2];
    if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) num_batches_tracked[0] += 1;   // nn.BatchNorm1d's counter, no launch of its own
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), js = threadIdx.x >> 4;
    double s0, s1;
    bn_sum_partials(C, nblocks, partial, c, js, red, s0, s1);
    if (js == 0 && c < C) {
        const double mu = s0 / (double)rows;
        double var = s1 / (double)rows - mu * mu;
        if (var < 0.0) var = 0.0;
        mean[c] = (float)mu;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
        if (running_var) {
            const double unbiased = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
}

// backward finalize: sums -> grad_weight / grad_bias and the two means used by the element pass (coef[0][c], coef[1][c])
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(long long rows, int C, int nblocks, const float* __restrict__ partial,
                                                              float* __restrict__ grad_weight, float* __restrict__ grad_bias, float* __restrict__ coef)
{
    __shared__ double red[16][16][2];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), js = threadIdx.x >> 4;
    double s0, s1;
    bn_sum_partials(C, nblocks, partial, c, js, red, s0, s1);
    if (js == 0 && c < C) {
        if (grad_bias) grad_bias[c] = (float)s0;
        if (grad_weight) grad_weight[c] = (float)s1;
        coef[c] = (float)(s0 / (double)rows);
        coef[C + c] = (float)(s1 / (double)rows);
    }
}

// MODE 0: y = [relu](xhat * w + b [+ residual]);   MODE 1: grad_x = w * invstd * (g - coef0 - xhat * coef1), grad_residual = g (the masked gradient)
template <int VEC, int MODE>
__global__ __launch_bounds__(BN_BLOCK) void bn_element_kernel(long long rows, int C, const float* __restrict__ x, const float* __restrict__ gy,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              const float* __restrict__ weight, const float* __restrict__ bias, const float* __restrict__ coef,
                                                              int relu, const float* __restrict__ residual, float* __restrict__ out, float* __restrict__ gres)
{
    const int tpr = C / VEC;
    const long long total = rows * tpr;
    const long long stride = (long long)gridDim.x * BN_BLOCK;
    // a lane's channels are the same on every trip when the grid stride is a multiple of the lanes per row (every power-of-two C): its per-channel
    // constants are then read ONCE (they were 4 .. 6 scalar loads per channel and trip, the bulk of the kernel's memory instructions) — same arithmetic
    const bool fixed = (stride % tpr) == 0;
    float cw[VEC], cb[VEC], cm[VEC], cis[VEC], k0[VEC], k1[VEC];
    auto constants = [&](int c0) {
#pragma unroll
        for (int v = 0; v < VEC; v++) {
            const int c = c0 + v;
            cw[v] = weight ? weight[c] : 1.f; cb[v] = bias ? bias[c] : 0.f; cm[v] = mean[c]; cis[v] = invstd[c];
            k0[v] = (MODE == 1) ? coef[c] : 0.f; k1[v] = (MODE == 1) ? coef[C + c] : 0.f;
        }
    };
    const long long e0 = (long long)blockIdx.x * BN_BLOCK + threadIdx.x;
    if (fixed && e0 < total) constants((int)(e0 % tpr) * VEC);
    for (long long e = e0; e < total; e += stride) {
        if (!fixed) constants((int)(e % tpr) * VEC);
        float xv[VEC], gv[VEC], o[VEC], rv[VEC], gm[VEC];
#pragma unroll
        for (int v = 0; v < VEC; v++) rv[v] = 0.f;
        if (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4*>(x + e * 4);
            xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
            if (MODE == 1) { const float4 u = *reinterpret_cast<const float4*>(gy + e * 4); gv[0] = u.x; gv[1] = u.y; gv[2] = u.z; gv[3] = u.w; }
            if (residual) { const float4 u = *reinterpret_cast<const float4*>(residual + e * 4); rv[0] = u.x; rv[1] = u.y; rv[2] = u.z; rv[3] = u.w; }
        } else {
            xv[0] = x[e];
            if (MODE == 1) gv[0] = gy[e];
            if (residual) rv[0] = residual[e];
        }
#pragma unroll
        for (int v = 0; v < VEC; v++) {
            const float xh = (xv[v] - cm[v]) * cis[v];
            const float y = (xh * cw[v] + cb[v]) + rv[v];
            if (MODE == 0) o[v] = (relu && !(y > 0.f)) ? 0.f : y;
            else {
                const float g = (relu && !(y > 0.f)) ? 0.f : gv[v];
                gm[v] = g;
                o[v] = cw[v] * cis[v] * ((g - k0[v]) - xh * k1[v]);
            }
        }
        if (VEC == 4) *reinterpret_cast<float4*>(out + e * 4) = make_float4(o[0], o[1], o[2], o[3]);
        else out[e] = o[0];
        if (MODE == 1 && gres) {
            if (VEC == 4) *reinterpret_cast<float4*>(gres + e * 4) = make_float4(gm[0], gm[1], gm[2], gm[3]);
            else gres[e] = gm[0];
        }
    }
}

// ---- small problems (rows <= 4096: the three coarse stages of the network, n = 2560 / 640 / 160): ONE kernel per direction.  The three-launch
// scheme above costs three graph nodes of 5-11 us each for tensors that fit a few registers per lane: here a workgroup owns 4 channels (one
// float4 column), lane t holds rows t, t + 256, ... (<= 16 of them) in registers, so x is read once, the statistics are reduced inside the
// workgroup (fp32 per wave, fp64 across waves, like the partial sums above) and the second pass runs from registers.
constexpr int BN_SMALL_ROWS = 4096, BN_SMALL_PER = BN_SMALL_ROWS / BN_BLOCK;

__device__ __forceinline__ float bn_wave_sum(float v) { return group_sum<64>(v); }    // DPP steps: six ds_bpermute round trips per value were most of the kernel

// sums[k] over the workgroup for k < NV, every thread gets the totals (as double)
template <int NV>
__device__ __forceinline__ void bn_block_sums(const float (&v)[NV], double (&tot)[NV], double (*red)[NV])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; k++) { const float w = bn_wave_sum(v[k]); if (lane == 0) red[wave][k] = (double)w; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; k++) tot[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    __syncthreads();
}

__global__ __launch_bounds__(BN_BLOCK) void bn_small_fwd_kernel(int rows, int C, const float* __restrict__ x, const float* __restrict__ weight,
                                                                const float* __restrict__ bias, float eps, float momentum,
                                                                float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                long long* __restrict__ num_batches_tracked, int relu, const float* __restrict__ residual,
                                                                float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ y)
{
    __shared__ double red[4][4];
    const int c0 = blockIdx.x * 4;
    if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) num_batches_tracked[0] += 1;
    float4 xv[BN_SMALL_PER];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < BN_SMALL_PER; k++) {
        const int r = k * BN_BLOCK + (int)threadIdx.x;
        xv[k] = r < rows ? *reinterpret_cast<const float4*>(x + (size_t)r * C + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        s[0] += xv[k].x; s[1] += xv[k].y; s[2] += xv[k].z; s[3] += xv[k].w;
    }
    // two passes over the rows (they sit in registers): the variance as the mean of squared deviations, not E[x^2] - mean^2, which loses every
    // digit once |mean| >> std (what torch's Welford update avoids as well)
    double tot[4], tot2[4];
    bn_block_sums<4>(s, tot, red);
    float mf[4];
#pragma unroll
    for (int v = 0; v < 4; v++) mf[v] = (float)(tot[v] / (double)rows);
    float s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < BN_SMALL_PER; k++) {
        const int r = k * BN_BLOCK + (int)threadIdx.x;
        if (r < rows) {
            const float dx = xv[k].x - mf[0], dy = xv[k].y - mf[1], dz = xv[k].z - mf[2], dw = xv[k].w - mf[3];
            s2[0] += dx * dx; s2[1] += dy * dy; s2[2] += dz * dz; s2[3] += dw * dw;
        }
    }
    bn_block_sums<4>(s2, tot2, red);
    float mu[4], is[4], w[4], b[4];
#pragma unroll
    for (int v = 0; v < 4; v++) {
        const double m = tot[v] / (double)rows;
        // sum (x - mf)^2 = sum (x - m)^2 + rows (m - mf)^2: the float rounding of the mean is taken back out
        double var = tot2[v] / (double)rows - (m - (double)mf[v]) * (m - (double)mf[v]);
        if (var < 0.0) var = 0.0;
        mu[v] = (float)m;
        // 1 / sqrt in fp32 with one Newton step on the hardware estimate (the fp64 divide + square root were ~250 instructions per workgroup)
        const float vf = (float)var + eps; float rs = __builtin_amdgcn_rsqf(vf); rs = rs * (1.5f - 0.5f * vf * rs * rs); is[v] = rs;
        w[v] = weight ? weight[c0 + v] : 1.f; b[v] = bias ? bias[c0 + v] : 0.f;
        if (threadIdx.x == 0) {
            mean[c0 + v] = mu[v]; invstd[c0 + v] = is[v];
            if (running_mean) running_mean[c0 + v] = (1.f - momentum) * running_mean[c0 + v] + momentum * mu[v];
            if (running_var) {
                const double unbiased = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
                running_var[c0 + v] = (1.f - momentum) * running_var[c0 + v] + momentum * (float)unbiased;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < BN_SMALL_PER; k++) {
        const int r = k * BN_BLOCK + (int)threadIdx.x;
        if (r < rows) {
            const float in[4] = {xv[k].x, xv[k].y, xv[k].z, xv[k].w};
            const float4 rr = residual ? *reinterpret_cast<const float4*>(residual + (size_t)r * C + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float rs4[4] = {rr.x, rr.y, rr.z, rr.w};
            float o[4];
#pragma unroll
            for (int v = 0; v < 4; v++) { const float yv = (((in[v] - mu[v]) * is[v]) * w[v] + b[v]) + rs4[v]; o[v] = (relu && !(yv > 0.f)) ? 0.f : yv; }
            *reinterpret_cast<float4*>(y + (size_t)r * C + c0) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

__global__ __launch_bounds__(BN_BLOCK) void bn_small_bwd_kernel(int rows, int C, const float* __restrict__ x, const float* __restrict__ gy,
                                                                const float* __restrict__ weight, const float* __restrict__ bias,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd, int relu,
                                                                const float* __restrict__ residual, float* __restrict__ gx, float* __restrict__ gres,
                                                                float* __restrict__ grad_weight, float* __restrict__ grad_bias)
{
    __shared__ double red[4][8];
    const int c0 = blockIdx.x * 4;
    float mu[4], is[4], w[4], b[4];
#pragma unroll
    for (int v = 0; v < 4; v++) { mu[v] = mean[c0 + v]; is[v] = invstd[c0 + v]; w[v] = weight ? weight[c0 + v] : 1.f; b[v] = bias ? bias[c0 + v] : 0.f; }
    float4 xh[BN_SMALL_PER], g[BN_SMALL_PER];                       // xhat and the (ReLU-masked) incoming gradient
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < BN_SMALL_PER; k++) {
        const int r = k * BN_BLOCK + (int)threadIdx.x;
        float a[4] = {0.f, 0.f, 0.f, 0.f}, gg[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < rows) {
            const float4 xr = *reinterpret_cast<const float4*>(x + (size_t)r * C + c0), gr = *reinterpret_cast<const float4*>(gy + (size_t)r * C + c0);
            const float4 rr = residual ? *reinterpret_cast<const float4*>(residual + (size_t)r * C + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float in[4] = {xr.x, xr.y, xr.z, xr.w}, gi[4] = {gr.x, gr.y, gr.z, gr.w}, rs4[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
            for (int v = 0; v < 4; v++) {
                a[v] = (in[v] - mu[v]) * is[v];
                const float yv = (a[v] * w[v] + b[v]) + rs4[v];
                gg[v] = (relu && !(yv > 0.f)) ? 0.f : gi[v];
            }
        }
        xh[k] = make_float4(a[0], a[1], a[2], a[3]); g[k] = make_float4(gg[0], gg[1], gg[2], gg[3]);
#pragma unroll
        for (int v = 0; v < 4; v++) { s[v] += gg[v]; s[4 + v] += gg[v] * a[v]; }
    }
    double tot[8];
    bn_block_sums<8>(s, tot, red);
    float k0[4], k1[4];
#pragma unroll
    for (int v = 0; v < 4; v++) {
        k0[v] = (float)(tot[v] / (double)rows); k1[v] = (float)(tot[4 + v] / (double)rows);
        if (threadIdx.x == 0) { if (grad_bias) grad_bias[c0 + v] = (float)tot[v]; if (grad_weight) grad_weight[c0 + v] = (float)tot[4 + v]; }
    }
#pragma unroll
    for (int k = 0; k < BN_SMALL_PER; k++) {
        const int r = k * BN_BLOCK + (int)threadIdx.x;
        if (r < rows) {
            const float a[4] = {xh[k].x, xh[k].y, xh[k].z, xh[k].w}, gg[4] = {g[k].x, g[k].y, g[k].z, g[k].w};
            float o[4];
#pragma unroll
            for (int v = 0; v < 4; v++) o[v] = w[v] * is[v] * ((gg[v] - k0[v]) - a[v] * k1[v]);
            *reinterpret_cast<float4*>(gx + (size_t)r * C + c0) = make_float4(o[0], o[1], o[2], o[3]);
            if (gres) *reinterpret_cast<float4*>(gres + (size_t)r * C + c0) = g[k];
        }
    }
}

inline bool bn_small(long long rows, int C) { return rows <= BN_SMALL_ROWS && C % 4 == 0; }

int bn_check(long long rows, int C)
{
    if (rows < 0 || C <= 0) return CBL_ERR_BAD_ARG;
    if (C > 1024 || (C % 4 != 0 && C > BN_BLOCK)) return CBL_ERR_UNSUPPORTED;
    return CBL_OK;
}

}  // namespace

// scratch: per-workgroup partial sums + (backward) the two coefficient rows
CBL_EXPORT size_t cbl_bn_rows_workspace_bytes(long long rows, int C)
{
    if (bn_check(rows, C) != CBL_OK) return 0;
    return sizeof(float) * ((size_t)BN_MAX_BLOCKS * 2 * C + 2 * (size_t)C) + 256;
}

CBL_EXPORT int cbl_bn_rows_forward_residual(long long rows, int C, const float* x, const float* residual, const float* weight, const float* bias, float eps,
                                            float momentum, float* running_mean, float* running_var, long long* num_batches_tracked, int relu, float* save_mean,
                                            float* save_invstd, float* y, void* workspace, size_t workspace_bytes, void* stream)
{
    const int rc = bn_check(rows, C);
    if (rc) return rc;
    if (rows == 0) return CBL_OK;
    if (!x || !save_mean || !save_invstd || !y || !workspace) return CBL_ERR_BAD_ARG;
    if (workspace_bytes < cbl_bn_rows_workspace_bytes(rows, C)) return CBL_ERR_WORKSPACE;
    const BnShape s = bn_shape(rows, C);
    float* partial = reinterpret_cast<float*>(workspace);
    hipStream_t st = cbl_stream(stream);
    if (bn_small(rows, C) && cbl_host_aligned16(x) && cbl_host_aligned16(y) && cbl_host_aligned16(residual)) {
        hipLaunchKernelGGL(bn_small_fwd_kernel, dim3(C / 4), dim3(BN_BLOCK), 0, st, (int)rows, C, x, weight, bias, eps, momentum, running_mean, running_var,
                           num_batches_tracked, relu, residual, save_mean, save_invstd, y);
        return cbl_status();
    }
    const bool vec = s.vec == 4 && cbl_host_aligned16(x) && cbl_host_aligned16(y) && cbl_host_aligned16(residual);
    const BnShape s1 = vec ? s : [&] { BnShape t = s; t.vec = 1; t.tpr = C; t.slots = BN_BLOCK / C; return t; }();
    if (s1.slots < 1) return CBL_ERR_UNSUPPORTED;
    if (vec) hipLaunchKernelGGL((bn_partial_kernel<4, 0>), dim3(s1.nblocks), dim3(BN_BLOCK), 0, st, rows, C, s1.tpr, s1.slots, s1.rows_per_block, x, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, partial);
    else     hipLaunchKernelGGL((bn_partial_kernel<1, 0>), dim3(s1.nblocks), dim3(BN_BLOCK), 0, st, rows, C, s1.tpr, s1.slots, s1.rows_per_block, x, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, partial);
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(cbl_div_up(C, 16)), dim3(256), 0, st, rows, C, s1.nblocks, partial, eps, momentum, running_mean, running_var, num_batches_tracked, save_mean, save_invstd);
    const dim3 grid(cbl_grid_for(rows * (C / (vec ? 4 : 1)), BN_BLOCK, 4096));
    if (vec) hipLaunchKernelGGL((bn_element_kernel<4, 0>), grid, dim3(BN_BLOCK), 0, st, rows, C, x, nullptr, save_mean, save_invstd, weight, bias, nullptr, relu, residual, y, nullptr);
    else     hipLaunchKernelGGL((bn_element_kernel<1, 0>), grid, dim3(BN_BLOCK), 0, st, rows, C, x, nullptr, save_mean, save_invstd, weight, bias, nullptr, relu, residual, y, nullptr);
    return cbl_status();
}

CBL_EXPORT int cbl_bn_rows_forward(long long rows, int C, const float* x, const float* weight, const float* bias, float eps, float momentum,
                                   float* running_mean, float* running_var, long long* num_batches_tracked, int relu, float* save_mean,
                                   float* save_invstd, float* y, void* workspace, size_t workspace_bytes, void* stream)
{
    return cbl_bn_rows_forward_residual(rows, C, x, nullptr, weight, bias, eps, momentum, running_mean, running_var, num_batches_tracked, relu, save_mean, save_invstd, y,
                                        workspace, workspace_bytes, stream);
}

CBL_EXPORT int cbl_bn_rows_backward_residual(long long rows, int C, const float* x, const float* residual, const float* grad_y, const float* weight, const float* bias,
                                             const float* save_mean, const float* save_invstd, int relu, float* grad_x, float* grad_residual, float* grad_weight,
                                             float* grad_bias, void* workspace, size_t workspace_bytes, void* stream)
{
    const int rc = bn_check(rows, C);
    if (rc) return rc;
    if (rows == 0) return CBL_OK;
    if (!x || !grad_y || !save_mean || !save_invstd || !grad_x || !workspace) return CBL_ERR_BAD_ARG;
    if (workspace_bytes < cbl_bn_rows_workspace_bytes(rows, C)) return CBL_ERR_WORKSPACE;
    const BnShape s = bn_shape(rows, C);
    float* partial = reinterpret_cast<float*>(workspace);
    float* coef = partial + (size_t)BN_MAX_BLOCKS * 2 * C;
    hipStream_t st = cbl_stream(stream);
    const bool al = cbl_host_aligned16(x) && cbl_host_aligned16(grad_y) && cbl_host_aligned16(grad_x) && cbl_host_aligned16(residual) && cbl_host_aligned16(grad_residual);
    if (bn_small(rows, C) && al) {
        hipLaunchKernelGGL(bn_small_bwd_kernel, dim3(C / 4), dim3(BN_BLOCK), 0, st, (int)rows, C, x, grad_y, weight, bias, save_mean, save_invstd, relu,
                           residual, grad_x, grad_residual, grad_weight, grad_bias);
        return cbl_status();
    }
    const bool vec = s.vec == 4 && al;
    const BnShape s1 = vec ? s : [&] { BnShape t = s; t.vec = 1; t.tpr = C; t.slots = BN_BLOCK / C; return t; }();
    if (s1.slots < 1) return CBL_ERR_UNSUPPORTED;
    if (vec) hipLaunchKernelGGL((bn_partial_kernel<4, 1>), dim3(s1.nblocks), dim3(BN_BLOCK), 0, st, rows, C, s1.tpr, s1.slots, s1.rows_per_block, x, grad_y, save_mean, save_invstd, weight, bias, relu, residual, partial);
    else     hipLaunchKernelGGL((bn_partial_kernel<1, 1>), dim3(s1.nblocks), dim3(BN_BLOCK), 0, st, rows, C, s1.tpr, s1.slots, s1.rows_per_block, x, grad_y, save_mean, save_invstd, weight, bias, relu, residual, partial);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cbl_div_up(C, 16)), dim3(256), 0, st, rows, C, s1.nblocks, partial, grad_weight, grad_bias, coef);
    const dim3 grid(cbl_grid_for(rows * (C / (vec ? 4 : 1)), BN_BLOCK, 4096));
    if (vec) hipLaunchKernelGGL((bn_element_kernel<4, 1>), grid, dim3(BN_BLOCK), 0, st, rows, C, x, grad_y, save_mean, save_invstd, weight, bias, coef, relu, residual, grad_x, grad_residual);
    else     hipLaunchKernelGGL((bn_element_kernel<1, 1>), grid, dim3(BN_BLOCK), 0, st, rows, C, x, grad_y, save_mean, save_invstd, weight, bias, coef, relu, residual, grad_x, grad_residual);
    return cbl_status();
}

CBL_EXPORT int cbl_bn_rows_backward(long long rows, int C, const float* x, const float* grad_y, const float* weight, const float* bias,
                                    const float* save_mean, const float* save_invstd, int relu, float* grad_x, float* grad_weight, float* grad_bias,
                                    void* workspace, size_t workspace_bytes, void* stream)
{
    return cbl_bn_rows_backward_residual(rows, C, x, nullptr, grad_y, weight, bias, save_mean, save_invstd, relu, grad_x, nullptr, grad_weight, grad_bias,
                                         workspace, workspace_bytes, stream);
}
