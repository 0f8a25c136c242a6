// a4: the C-wide part of the vector attention of PointTransformerLayer, without its (n, K, C) tensors.
// Reference: /root/reference/pytorch/model/blocks.py:31-44
//     p_r = linear_p(p_j - p_i)                      (n,K,C)      linear_p = Linear(3,3) BN ReLU | Linear(3,C)
//     w   = x_k[j] - x_q[i] + p_r                    (n,K,C)
//     w   = linear_w(w)                              (n,K,C/8)    linear_w = BN(C) ReLU Linear(C,C/8) | BN ReLU Linear(C/8,C/8)
//     out = sum_j (x_v[j] + p_r) * softmax_j(w)      (n,C)
// Everything left of the bars above is C wide and exists per (point, neighbour) pair; the reference (and the unfused mirror)
// materialises p_r, w, BN(w), ReLU(BN(w)) and their gradients: ~20 passes over 42 MB tensors per layer at (10240,16,64).
// Here the pair values live in registers and are RECOMPUTED by every pass that needs them; only the narrow tensors exist:
//     p1 = ReLU(BN(Linear(3,3)(p_j - p_i)))  (n,K,3)   in          w2 = Linear(C,C/8)(ReLU(BN(w)))  (n,K,C/8)   out
// and the train-mode BatchNorm(C) costs one extra statistics pass:
//   attn_w2 forward   P1  per-channel partial sums of w and w^2 over all pairs      -> finalize (mean, invstd, running stats)
//                     P2  w -> BN -> ReLU -> the C x C/8 product, reduced across the C lanes of the pair's group
//   attn_w2 backward  Q1  recompute, d(ReLU(BN(w))) from grad_w2; BatchNorm's two reductions; grad of Linear(C,C/8)'s parameters
//                     Q2  recompute, grad_w by the BatchNorm backward formula; scatter to x_k (atomics) / x_q; grads of
//                         Linear(3,C)'s parameters and of p1 (3 group reductions per pair)
//   attn_agg forward / backward: K9 / K10 with p_r computed on the fly from p1 (and its parameter / p1 gradients in the backward)
// One group of C lanes (C = 32 or 64: the two full-resolution stages, where the big tensors are) owns one point and walks its K
// neighbours; x_k / x_v rows are read as coalesced 4*C-byte segments; parameter gradients are accumulated per lane in registers
// over a persistent group's points, combined per workgroup in LDS, written as per-workgroup partials and summed in fp64.
#include "cbl_common.h"
#include <stdlib.h>

namespace {

constexpr int AT_BLOCK = 256;
constexpr int AT_MAX_BLOCKS = 1024;
constexpr int AT_U = 4;                     // pairs whose loads are in flight together
constexpr int AT_MAXG = 8;                  // C / share_planes with C <= 64

// sum over the C lanes of a group, result in every lane of the group (C = 32: two groups per wave)
// (one v_add_f32_dpp per step: the move folds into the add with old = 0 + bound_ctrl; the lanes read afterwards receive the same terms in the same order as with
//  row-masked moves — wave_ops.h group_sum)
template <int CTRL> __device__ __forceinline__ float dpp_add0(float v) { return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); }
template <int C> __device__ __forceinline__ float group_sum(float v)
{
    v = dpp_add0<0xB1>(v); v = dpp_add0<0x4E>(v); v = dpp_add0<0x141>(v); v = dpp_add0<0x140>(v);
    v = dpp_add0<0x142>(v);
    if (C == 32) return __shfl(v, 31, 32);
    v = dpp_add0<0x143>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

struct PairParams {          // per-lane (= per-channel) constants of the C-wide chain
    float w0, w1, w2, b;     // Linear(3, C): row c of the weight, bias
    float mean, invstd, gamma, beta;
};

__device__ __forceinline__ float pe_of(const PairParams& q, float a0, float a1, float a2) { return ((q.b + a0 * q.w0) + a1 * q.w1) + a2 * q.w2; }


// the U next pairs of point i: neighbour ids, the three p1 values and this lane's x row element of each — all loads issued before any use
struct PairBatch { int j[AT_U]; float a0[AT_U], a1[AT_U], a2[AT_U], xr[AT_U]; };
template <int C>
__device__ __forceinline__ void load_pairs(PairBatch& pb, int i, int k0, int K, int c, const int* __restrict__ idx, const float* __restrict__ p1,
                                           const float* __restrict__ rows)
{
#pragma unroll
    for (int u = 0; u < AT_U; u++) {
        const int k = min(k0 + u, K - 1);                            // tail: repeat the last pair (its result is not used)
        const size_t r = (size_t)i * K + k;
        pb.j[u] = idx[r];
        pb.a0[u] = p1[3 * r]; pb.a1[u] = p1[3 * r + 1]; pb.a2[u] = p1[3 * r + 2];
    }
#pragma unroll
    for (int u = 0; u < AT_U; u++) pb.xr[u] = rows[(size_t)pb.j[u] * C + c];
}

// ----------------------------------------------------------------------------------------------------------- attn_w2, P1
template <int C>
__global__ __launch_bounds__(AT_BLOCK) void attn_w2_stats_kernel(int n, int K, const float* __restrict__ xq, const float* __restrict__ xk,
                                                                 const int* __restrict__ idx, const float* __restrict__ p1,
                                                                 const float* __restrict__ W3C, const float* __restrict__ b3C, float* __restrict__ partial)
{
    constexpr int GPB = AT_BLOCK / C;                                // groups per workgroup
    __shared__ float red[2][AT_BLOCK];
    const int c = threadIdx.x % C, grp = threadIdx.x / C;
    PairParams q; q.w0 = W3C[3 * c]; q.w1 = W3C[3 * c + 1]; q.w2 = W3C[3 * c + 2]; q.b = b3C[c];
    float s0 = 0.f, s1 = 0.f;
    for (int i = blockIdx.x * GPB + grp; i < n; i += gridDim.x * GPB) {
        const float xqi = xq[(size_t)i * C + c];
        for (int k0 = 0; k0 < K; k0 += AT_U) {
            PairBatch pb; load_pairs<C>(pb, i, k0, K, c, idx, p1, xk);
#pragma unroll
            for (int u = 0; u < AT_U; u++) {
                if (k0 + u < K) {
                    const float w = pe_of(q, pb.a0[u], pb.a1[u], pb.a2[u]) - (xqi - pb.xr[u]);
                    s0 += w; s1 += w * w;
                }
            }
        }
    }
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
    __syncthreads();
    if (grp == 0) {
        double a0 = 0.0, a1 = 0.0;
        for (int g = 0; g < GPB; g++) { a0 += (double)red[0][g * C + c]; a1 += (double)red[1][g * C + c]; }
        partial[((size_t)blockIdx.x * 2) * C + c] = (float)a0;
        partial[((size_t)blockIdx.x * 2 + 1) * C + c] = (float)a1;
    }
}

// mean / invstd from the partials (+ running statistics, batch counter), 16 channels x 16 slices per workgroup
__global__ __launch_bounds__(256) void attn_bn_finalize_kernel(long long rows, int C, int nblocks, const float* __restrict__ partial, float eps, float momentum,
                                                               float* __restrict__ running_mean, float* __restrict__ running_var,
                                                               long long* __restrict__ num_batches_tracked, float* __restrict__ mean, float* __restrict__ invstd)
{
    __shared__ double red[16][16][2];
    if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) num_batches_tracked[0] += 1;
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), js = threadIdx.x >> 4;
    double a0 = 0.0, a1 = 0.0;
    if (c < C) {
#pragma unroll 8
        for (int b = js; b < nblocks; b += 16) { a0 += (double)partial[((size_t)b * 2) * C + c]; a1 += (double)partial[((size_t)b * 2 + 1) * C + c]; }
    }
    red[js][threadIdx.x & 15][0] = a0; red[js][threadIdx.x & 15][1] = a1;
    __syncthreads();
    if (js == 0 && c < C) {
        double s0 = 0.0, s1 = 0.0;
        for (int j = 0; j < 16; j++) { s0 += red[j][threadIdx.x & 15][0]; s1 += red[j][threadIdx.x & 15][1]; }
        const double mu = s0 / (double)rows;
        double var = s1 / (double)rows - mu * mu;
        if (var < 0.0) var = 0.0;
        mean[c] = (float)mu;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
        if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(rows > 1 ? var * (double)rows / (double)(rows - 1) : var);
    }
}

// ----------------------------------------------------------------------------------------------------------- attn_w2, P2
template <int C, int G>
__global__ __launch_bounds__(AT_BLOCK) void attn_w2_forward_kernel(int n, int K, const float* __restrict__ xq, const float* __restrict__ xk,
                                                                   const int* __restrict__ idx, const float* __restrict__ p1,
                                                                   const float* __restrict__ W3C, const float* __restrict__ b3C,
                                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   const float* __restrict__ Wa, const float* __restrict__ ba, float* __restrict__ w2)
{
    constexpr int GPB = AT_BLOCK / C;
    const int c = threadIdx.x % C, grp = threadIdx.x / C;
    PairParams q; q.w0 = W3C[3 * c]; q.w1 = W3C[3 * c + 1]; q.w2 = W3C[3 * c + 2]; q.b = b3C[c];
    q.mean = mean[c]; q.invstd = invstd[c]; q.gamma = gamma ? gamma[c] : 1.f; q.beta = beta ? beta[c] : 0.f;
    float wa[G];
#pragma unroll
    for (int g = 0; g < G; g++) wa[g] = Wa[g * C + c];
    const float bias_g = (c < G) ? ba[c] : 0.f;
    for (int i = blockIdx.x * GPB + grp; i < n; i += gridDim.x * GPB) {
        const float xqi = xq[(size_t)i * C + c];
        for (int k0 = 0; k0 < K; k0 += AT_U) {
            PairBatch pb; load_pairs<C>(pb, i, k0, K, c, idx, p1, xk);
#pragma unroll
            for (int u = 0; u < AT_U; u++) {
                if (k0 + u < K) {                                    // group-uniform
                    const size_t r = (size_t)i * K + k0 + u;
                    const float w = pe_of(q, pb.a0[u], pb.a1[u], pb.a2[u]) - (xqi - pb.xr[u]);
                    const float y = (w - q.mean) * q.invstd * q.gamma + q.beta;
                    const float w1 = y > 0.f ? y : 0.f;
                    float mine = 0.f;
#pragma unroll
                    for (int g = 0; g < G; g++) { const float t = group_sum<C>(wa[g] * w1); mine = (c == g) ? t : mine; }
                    if (c < G) w2[r * G + c] = mine + bias_g;
                }
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------------------- attn_w2, P2 on the matrix cores
// C = 32 / 64.  The kernel above reduces every output of the C x C/8 product across the C lanes of a pair with DPP adds (8 group sums of ~10
// instructions per pair: 233 us at (40960, 16, 64), bound by the vector ALU).  Here ONE WAVE owns a tile of 16 pairs of a point — the pairs are the
// M rows of v_mfma_f32_16x16x4_f32 — and lane (pair = l % 16, quarter = l / 16) owns the CONTIGUOUS quarter of the pair's channels: its x_k
// segment is C/4 floats of one row (16-byte loads), the per-channel constants are read from LDS (every lane of a quarter reads the same words: broadcast),
// step t of the contraction feeds y[pair][quarter C/4 + t] as the A operand against Wa[g][quarter C/4 + t] (B operand, in registers); the
// (16 pairs x C/8) result leaves the accumulator as rows of w2.  K = 8 (stage 0): a tile holds the 8 pairs of TWO points.
using at_f32x4 = __attribute__((ext_vector_type(4))) float;

template <int C>
__global__ __launch_bounds__(AT_BLOCK) void attn_w2_forward_mfma_kernel(int n, int K, const float* __restrict__ xq, const float* __restrict__ xk,
                                                                        const int* __restrict__ idx, const float* __restrict__ p1,
                                                                        const float* __restrict__ W3C, const float* __restrict__ b3C,
                                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                        const float* __restrict__ Wa, const float* __restrict__ ba, float* __restrict__ w2)
{
    constexpr int G = C / 8, CH = C / 4;
    // per-channel constants in LDS (all lanes of a quarter read the same words: broadcast): {w0, w1, w2, b}, {mean, invstd, gamma, beta}
    __shared__ float4 prm[C][2];
    const int lane = threadIdx.x & 63, pr = lane & 15, kq = lane >> 4;
    for (int c = threadIdx.x; c < C; c += AT_BLOCK) {
        prm[c][0] = make_float4(W3C[3 * c], W3C[3 * c + 1], W3C[3 * c + 2], b3C[c]);
        prm[c][1] = make_float4(mean[c], invstd[c], gamma ? gamma[c] : 1.f, beta ? beta[c] : 0.f);
    }
    __syncthreads();
    float bw[CH];
#pragma unroll
    for (int t = 0; t < CH; t++) bw[t] = pr < G ? Wa[(size_t)pr * C + CH * kq + t] : 0.f;      // B[k][j]: lane = j + 16 k
    const float bias_g = pr < G ? ba[pr] : 0.f;
    // a tile = 16 consecutive (point, neighbour) pairs of the flat (n * K) list when K divides 16 (K = 8: two points), else 16 pairs of one point
    const bool flat = (16 % K) == 0;
    const int tiles_per_point = flat ? 1 : (K + 15) / 16;
    const long long ntiles = flat ? ((long long)n * K + 15) / 16 : (long long)n * tiles_per_point;
    const long long npairs = (long long)n * K;
    for (long long tile = (long long)blockIdx.x * (AT_BLOCK / 64) + (threadIdx.x >> 6); tile < ntiles; tile += (long long)gridDim.x * (AT_BLOCK / 64)) {
        // this lane's pair (the A row pr) and the four rows 4 kq + r it will write
        long long pair; bool valid;
        if (flat) { pair = tile * 16 + pr; valid = pair < npairs; }
        else { const long long i = tile / tiles_per_point; const int k = (int)(tile - i * tiles_per_point) * 16 + pr; valid = k < K; pair = i * K + min(k, K - 1); }
        const long long pc = valid ? pair : npairs - 1;
        const long long i = pc / K;
        const int j = idx[pc];
        const float a0 = p1[3 * pc], a1 = p1[3 * pc + 1], a2 = p1[3 * pc + 2];
        const float4* xkr = reinterpret_cast<const float4*>(xk + (size_t)j * C + CH * kq);
        const float4* xqr = reinterpret_cast<const float4*>(xq + (size_t)i * C + CH * kq);
        float4 kv[CH / 4], qv[CH / 4];
#pragma unroll
        for (int v = 0; v < CH / 4; v++) { kv[v] = xkr[v]; qv[v] = xqr[v]; }
        at_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int v = 0; v < CH / 4; v++) {
            const float kx[4] = {kv[v].x, kv[v].y, kv[v].z, kv[v].w}, qx[4] = {qv[v].x, qv[v].y, qv[v].z, qv[v].w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int t = 4 * v + e;
                const float4 pa = prm[CH * kq + t][0], pb = prm[CH * kq + t][1];
                // the arithmetic of the other passes, operation for operation (the ReLU mask must be the same bit pattern in all of them)
                const float w = (((pa.w + a0 * pa.x) + a1 * pa.y) + a2 * pa.z) - (qx[e] - kx[e]);
                const float y = (w - pb.x) * pb.y * pb.z + pb.w;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(y > 0.f ? y : 0.f, bw[t], acc, 0, 0, 0);
            }
        }
        // D[4 (lane / 16) + r][lane % 16] = w2 of pair row 4 kq + r, column g = pr
        if (pr < G) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 4 * kq + r;
                long long op; bool ok;
                if (flat) { op = tile * 16 + row; ok = op < npairs; }
                else { const long long i2 = tile / tiles_per_point; const int k2 = (int)(tile - i2 * tiles_per_point) * 16 + row; ok = k2 < K; op = i2 * K + k2; }
                if (ok) w2[op * G + pr] = acc[r] + bias_g;
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------------------- attn_w2, Q1 / Q2
// per-workgroup partial layout (floats): [0,C) S0 | [C,2C) S1 | [2C, 2C+G*C) dWa | [.., +G) dba          (Q1)
//                                         [0,3C) dW3C | [3C,4C) db3C                                        (Q2, agg backward)
template <int C, int G>
__global__ __launch_bounds__(AT_BLOCK) void attn_w2_bwd_reduce_kernel(int n, int K, const float* __restrict__ xq, const float* __restrict__ xk,
                                                                      const int* __restrict__ idx, const float* __restrict__ p1,
                                                                      const float* __restrict__ W3C, const float* __restrict__ b3C,
                                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                      const float* __restrict__ Wa, const float* __restrict__ gw2, float* __restrict__ partial)
{
    constexpr int GPB = AT_BLOCK / C;
    constexpr int NV = 2 + G + 1;                                    // values per lane: S0, S1, dWa[0..G), dba (lane g < G)
    __shared__ float red[NV][AT_BLOCK];
    const int c = threadIdx.x % C, grp = threadIdx.x / C;
    PairParams q; q.w0 = W3C[3 * c]; q.w1 = W3C[3 * c + 1]; q.w2 = W3C[3 * c + 2]; q.b = b3C[c];
    q.mean = mean[c]; q.invstd = invstd[c]; q.gamma = gamma ? gamma[c] : 1.f; q.beta = beta ? beta[c] : 0.f;
    float wa[G], dwa[G];
#pragma unroll
    for (int g = 0; g < G; g++) { wa[g] = Wa[g * C + c]; dwa[g] = 0.f; }
    float s0 = 0.f, s1 = 0.f, dba = 0.f;
    for (int i = blockIdx.x * GPB + grp; i < n; i += gridDim.x * GPB) {
        const float xqi = xq[(size_t)i * C + c];
        for (int k0 = 0; k0 < K; k0 += AT_U) {
            PairBatch pb; load_pairs<C>(pb, i, k0, K, c, idx, p1, xk);
            float gd[AT_U][G];
#pragma unroll
            for (int u = 0; u < AT_U; u++) {
                const size_t r = (size_t)i * K + min(k0 + u, K - 1);
#pragma unroll
                for (int g = 0; g < G; g++) gd[u][g] = gw2[r * G + g];
            }
#pragma unroll
            for (int u = 0; u < AT_U; u++) {
                if (k0 + u < K) {
                    const float w = pe_of(q, pb.a0[u], pb.a1[u], pb.a2[u]) - (xqi - pb.xr[u]);
                    const float xh = (w - q.mean) * q.invstd;
                    const float y = xh * q.gamma + q.beta;
                    const float w1 = y > 0.f ? y : 0.f;
                    float dw1 = 0.f;
#pragma unroll
                    for (int g = 0; g < G; g++) { dw1 += wa[g] * gd[u][g]; dwa[g] += gd[u][g] * w1; }
                    const float dy = y > 0.f ? dw1 : 0.f;
                    s0 += dy; s1 += dy * xh;
#pragma unroll
                    for (int g = 0; g < G; g++) dba += (c == g) ? gd[u][g] : 0.f;
                }
            }
        }
    }
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
#pragma unroll
    for (int g = 0; g < G; g++) red[2 + g][threadIdx.x] = dwa[g];
    red[2 + G][threadIdx.x] = dba;
    __syncthreads();
    if (grp == 0) {
        float* mine = partial + (size_t)blockIdx.x * (2 * C + G * C + G);
        float t[NV];
#pragma unroll
        for (int v = 0; v < NV; v++) { float a = 0.f; for (int g2 = 0; g2 < GPB; g2++) a += red[v][g2 * C + c]; t[v] = a; }
        mine[c] = t[0]; mine[C + c] = t[1];
#pragma unroll
        for (int g = 0; g < G; g++) mine[2 * C + g * C + c] = t[2 + g];
        if (c < G) mine[2 * C + G * C + c] = t[2 + G];
    }
}

// sum_b partial[b][e] (fp64) for e < nvals, 16 values x 16 slices per workgroup; the values go to up to four destination arrays laid
// end to end (segment s covers [seg.begin[s], seg.begin[s+1])), and — if `sums` is given — also to sums[e]
struct SumSegments { float* dst[4]; int begin[5]; };
__global__ __launch_bounds__(256) void attn_sum_partials_kernel(int nvals, int nblocks, const float* __restrict__ partial, SumSegments seg, float* __restrict__ sums)
{
    __shared__ double red[16][16];
    const int e = blockIdx.x * 16 + (threadIdx.x & 15), js = threadIdx.x >> 4;
    double a = 0.0;
    if (e < nvals) {
#pragma unroll 8
        for (int b = js; b < nblocks; b += 16) a += (double)partial[(size_t)b * nvals + e];
    }
    red[js][threadIdx.x & 15] = a;
    __syncthreads();
    if (js == 0 && e < nvals) {
        double s = 0.0;
        for (int j = 0; j < 16; j++) s += red[j][threadIdx.x & 15];
        if (sums) sums[e] = (float)s;
#pragma unroll
        for (int t = 0; t < 4; t++)
            if (seg.dst[t] && e >= seg.begin[t] && e < seg.begin[t + 1]) seg.dst[t][e - seg.begin[t]] = (float)s;
    }
}

// Q2: sums[0..C) = S0 = grad_beta, sums[C..2C) = S1 = grad_gamma (from Q1's finalize)
template <int C, int G>
__global__ __launch_bounds__(AT_BLOCK) void attn_w2_bwd_apply_kernel(int n, int K, const float* __restrict__ xq, const float* __restrict__ xk,
                                                                     const int* __restrict__ idx, const float* __restrict__ p1,
                                                                     const float* __restrict__ W3C, const float* __restrict__ b3C,
                                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                     const float* __restrict__ Wa, const float* __restrict__ gw2, const float* __restrict__ sums,
                                                                     float* __restrict__ gxq, float* __restrict__ gxk, float* __restrict__ gp1,
                                                                     float* __restrict__ partial)
{
    constexpr int GPB = AT_BLOCK / C;
    __shared__ float red[4][AT_BLOCK];
    const int c = threadIdx.x % C, grp = threadIdx.x / C;
    PairParams q; q.w0 = W3C[3 * c]; q.w1 = W3C[3 * c + 1]; q.w2 = W3C[3 * c + 2]; q.b = b3C[c];
    q.mean = mean[c]; q.invstd = invstd[c]; q.gamma = gamma ? gamma[c] : 1.f; q.beta = beta ? beta[c] : 0.f;
    float wa[G];
#pragma unroll
    for (int g = 0; g < G; g++) wa[g] = Wa[g * C + c];
    const float inv_rows = 1.0f / ((float)n * (float)K);
    const float c0 = sums[c] * inv_rows, c1 = sums[C + c] * inv_rows;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, db = 0.f;                    // dW3C[c, 0..2], db3C[c]
    for (int i = blockIdx.x * GPB + grp; i < n; i += gridDim.x * GPB) {
        const float xqi = xq[(size_t)i * C + c];
        float gq = 0.f;
        for (int k0 = 0; k0 < K; k0 += AT_U) {
            PairBatch pb; load_pairs<C>(pb, i, k0, K, c, idx, p1, xk);
            float gd[AT_U][G];
#pragma unroll
            for (int u = 0; u < AT_U; u++) {
                const size_t r = (size_t)i * K + min(k0 + u, K - 1);
#pragma unroll
                for (int g = 0; g < G; g++) gd[u][g] = gw2[r * G + g];
            }
#pragma unroll
            for (int u = 0; u < AT_U; u++) {
                if (k0 + u < K) {
                    const size_t r = (size_t)i * K + k0 + u;
                    const float a0 = pb.a0[u], a1 = pb.a1[u], a2 = pb.a2[u];
                    const float w = pe_of(q, a0, a1, a2) - (xqi - pb.xr[u]);
                    const float xh = (w - q.mean) * q.invstd;
                    const float y = xh * q.gamma + q.beta;
                    float dw1 = 0.f;
#pragma unroll
                    for (int g = 0; g < G; g++) dw1 += wa[g] * gd[u][g];
                    const float dy = y > 0.f ? dw1 : 0.f;
                    const float dw = q.gamma * q.invstd * ((dy - c0) - xh * c1);  // BatchNorm backward, train mode
                    if (gxk) unsafeAtomicAdd(gxk + (size_t)pb.j[u] * C + c, dw);     // w = p_r - x_q + x_k[j]  (NULL: gathered by attn_w2_gxk_csr_kernel)
                    gq -= dw;
                    d0 += dw * a0; d1 += dw * a1; d2 += dw * a2; db += dw;
                    const float t0 = group_sum<C>(q.w0 * dw), t1 = group_sum<C>(q.w1 * dw), t2 = group_sum<C>(q.w2 * dw);
                    if (c < 3) gp1[3 * r + c] = (c == 0) ? t0 : (c == 1) ? t1 : t2;
                }
            }
        }
        gxq[(size_t)i * C + c] = gq;
    }
    red[0][threadIdx.x] = d0; red[1][threadIdx.x] = d1; red[2][threadIdx.x] = d2; red[3][threadIdx.x] = db;
    __syncthreads();
    if (grp == 0) {
        float* mine = partial + (size_t)blockIdx.x * (4 * C);
        float t[4];
#pragma unroll
        for (int v = 0; v < 4; v++) { float a = 0.f; for (int g2 = 0; g2 < GPB; g2++) a += red[v][g2 * C + c]; t[v] = a; }
        mine[3 * c] = t[0]; mine[3 * c + 1] = t[1]; mine[3 * c + 2] = t[2]; mine[3 * C + c] = t[3];
    }
}

// ----------------------------------------------------------------------------------------------------------- attn_agg
// out[i,c] = sum_k (x_v[j,c] + p_r[i,k,c]) * a[i,k,c % G]                       blocks.py:42-43 (K9 with p_r on the fly)
// softmax over the K neighbours of point i for group column g (blocks.py:41), fused into the aggregation: max and sum from the logits
// (K reads each, L1 hits), the weights themselves are formed where they are used
struct SoftmaxCol { float m, inv; };
__device__ __forceinline__ SoftmaxCol softmax_col(const float* __restrict__ lg, int i, int K, int G, int g)
{
    const float* col = lg + (size_t)i * K * G + g;
    float m = -INFINITY;
    for (int k = 0; k < K; k++) m = fmaxf(m, col[(size_t)k * G]);
    float ssum = 0.f;
    for (int k = 0; k < K; k++) ssum += __builtin_amdgcn_exp2f((col[(size_t)k * G] - m) * 1.44269504088896340736f);
    SoftmaxCol r; r.m = m; r.inv = 1.0f / ssum;
    return r;
}
__device__ __forceinline__ float softmax_w(const SoftmaxCol& sc, float logit) { return __builtin_amdgcn_exp2f((logit - sc.m) * 1.44269504088896340736f) * sc.inv; }

template <int C, int G>
__global__ __launch_bounds__(AT_BLOCK) void attn_agg_forward_kernel(int n, int K, const float* __restrict__ xv, const int* __restrict__ idx,
                                                                    const float* __restrict__ p1, const float* __restrict__ W3C, const float* __restrict__ b3C,
                                                                    const float* __restrict__ a, float* __restrict__ a_out, float* __restrict__ out)
{
    // a_out != nullptr: `a` holds the LOGITS; the softmax over K is applied here and its result written to a_out for the backward pass
    constexpr int GPB = AT_BLOCK / C;
    const int c = threadIdx.x % C, grp = threadIdx.x / C;
    PairParams q; q.w0 = W3C[3 * c]; q.w1 = W3C[3 * c + 1]; q.w2 = W3C[3 * c + 2]; q.b = b3C[c];
    for (int i = blockIdx.x * GPB + grp; i < n; i += gridDim.x * GPB) {
        SoftmaxCol sc = {0.f, 1.f};
        if (a_out) sc = softmax_col(a, i, K, G, c % G);
        float acc = 0.f;
        for (int k0 = 0; k0 < K; k0 += AT_U) {
            PairBatch pb; load_pairs<C>(pb, i, k0, K, c, idx, p1, xv);
            float av[AT_U];
#pragma unroll
            for (int u = 0; u < AT_U; u++) {
                av[u] = a[((size_t)i * K + min(k0 + u, K - 1)) * G + (c % G)];
                if (a_out) { av[u] = softmax_w(sc, av[u]); if (c < G && k0 + u < K) a_out[((size_t)i * K + k0 + u) * G + c] = av[u]; }
            }
#pragma unroll
            for (int u = 0; u < AT_U; u++)
                if (k0 + u < K) acc += (pb.xr[u] + pe_of(q, pb.a0[u], pb.a1[u], pb.a2[u])) * av[u];
        }
        out[(size_t)i * C + c] = acc;
    }
}

// sum over the first G lanes of a 16-lane row (G = 4, 8, 16): every one of them ends up with the total
template <int G> __device__ __forceinline__ float low_lanes_sum(float v)
{
    static_assert(G == 4 || G == 8 || G == 16, "group width");
    v = dpp_add0<0xB1>(v);                                           // quad_perm [1,0,3,2]
    v = dpp_add0<0x4E>(v);                                           // quad_perm [2,3,0,1]
    if (G >= 8) v = dpp_add0<0x141>(v);                         // row_half_mirror
    if (G >= 16) v = dpp_add0<0x140>(v);                        // row_mirror
    return v;
}

// Per pair the kernel needs two contractions of the point's output gradient g[c] = grad_out[i, c]:
//   grad_p1[pair, a] = sum_c w_a[c] g[c] a[pair, c % G]          = sum_g a[pair, g] T_a[g]
//   grad_a[pair, g]  = sum_{c = g mod G} g[c] (x_v[j, c] + pe_c)  = sum_{c = g mod G} g[c] x_v[j, c]  +  B[g] + p1_0 T_0[g] + p1_1 T_1[g] + p1_2 T_2[g]
// with T_a[g] = sum_{c = g mod G} w_a[c] g[c] and B[g] = sum_{c = g mod G} b[c] g[c] — functions of the POINT, formed once per point (round 2 ran three
// 64-lane reductions per pair for grad_p1: 3 x 16 per point): per pair one strided sum for the gathered row and a G-lane sum for grad_p1 remain.
template <int C, int G>
__global__ __launch_bounds__(AT_BLOCK) void attn_agg_backward_kernel(int n, int K, const float* __restrict__ xv, const int* __restrict__ idx,
                                                                     const float* __restrict__ p1, const float* __restrict__ W3C, const float* __restrict__ b3C,
                                                                     const float* __restrict__ a, const float* __restrict__ go,
                                                                     float* __restrict__ gxv, float* __restrict__ gp1, float* __restrict__ ga, float* __restrict__ partial,
                                                                     int softmax)
{
    // softmax: `a` is the softmax output saved by the forward pass and `ga` receives the gradient of the LOGITS,
    // a[k] (da[k] - sum_k' a[k'] da[k']): the lanes c < G of a point's group hold column g = c and finish it when the point's pairs are done
    constexpr int GPB = AT_BLOCK / C;
    __shared__ float red[4][AT_BLOCK];
    const int c = threadIdx.x % C, grp = threadIdx.x / C;
    PairParams q; q.w0 = W3C[3 * c]; q.w1 = W3C[3 * c + 1]; q.w2 = W3C[3 * c + 2]; q.b = b3C[c];
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, db = 0.f;
    for (int i = blockIdx.x * GPB + grp; i < n; i += gridDim.x * GPB) {
        const float g = go[(size_t)i * C + c];
        float T0 = q.w0 * g, T1 = q.w1 * g, T2 = q.w2 * g, Bg = q.b * g;      // -> sums over this lane's residue class c mod G (every lane of the class holds them)
#pragma unroll
        for (int st = G; st < C; st <<= 1) { T0 += __shfl_xor(T0, st); T1 += __shfl_xor(T1, st); T2 += __shfl_xor(T2, st); Bg += __shfl_xor(Bg, st); }
        float dot = 0.f;
        for (int k0 = 0; k0 < K; k0 += AT_U) {
          PairBatch pb; load_pairs<C>(pb, i, k0, K, c, idx, p1, xv);
          float avs[AT_U];
#pragma unroll
          for (int u = 0; u < AT_U; u++) avs[u] = a[((size_t)i * K + min(k0 + u, K - 1)) * G + (c % G)];
#pragma unroll
          for (int u = 0; u < AT_U; u++) {
            if (k0 + u >= K) continue;
            const size_t r = (size_t)i * K + k0 + u;
            const int j = pb.j[u];
            const float a0 = pb.a0[u], a1 = pb.a1[u], a2 = pb.a2[u];
            const float av = avs[u];
            const float dpe = g * av;                                // d out / d (x_v[j] + p_r)
            if (gxv) unsafeAtomicAdd(gxv + (size_t)j * C + c, dpe);     // NULL: gathered by attn_agg_gxv_csr_kernel
            d0 += dpe * a0; d1 += dpe * a1; d2 += dpe * a2; db += dpe;
            // grad_p1: the lanes c < G hold a[pair, c] and T_a[c]
            const float t0 = low_lanes_sum<G>(av * T0), t1 = low_lanes_sum<G>(av * T1), t2 = low_lanes_sum<G>(av * T2);
            if (c < 3) gp1[3 * r + c] = (c == 0) ? t0 : (c == 1) ? t1 : t2;
            // grad_a[i,k,g]: the gathered row's part over the C/G lanes at stride G, the positional part from the point's sums
            float da = g * pb.xr[u];
#pragma unroll
            for (int st = G; st < C; st <<= 1) da += __shfl_xor(da, st);
            da += ((Bg + a0 * T0) + a1 * T1) + a2 * T2;
            if (c < G) { ga[r * G + c] = da; dot += av * da; }
          }
        }
        if (softmax && c < G)                                        // this lane wrote ga[.., c] itself: reads see its own stores
            for (int k = 0; k < K; k++) { const size_t r = (size_t)i * K + k; ga[r * G + c] = a[r * G + c] * (ga[r * G + c] - dot); }
    }
    red[0][threadIdx.x] = d0; red[1][threadIdx.x] = d1; red[2][threadIdx.x] = d2; red[3][threadIdx.x] = db;
    __syncthreads();
    if (grp == 0) {
        float* mine = partial + (size_t)blockIdx.x * (4 * C);
        float t[4];
#pragma unroll
        for (int v = 0; v < 4; v++) { float s = 0.f; for (int g2 = 0; g2 < GPB; g2++) s += red[v][g2 * C + c]; t[v] = s; }
        mine[3 * c] = t[0]; mine[3 * c + 1] = t[1]; mine[3 * c + 2] = t[2]; mine[3 * C + c] = t[3];
    }
}

// ----------------------------------------------------------------------------------------------------------- the two scatters as gathers
// d loss / d x_k[j] and d loss / d x_v[j] are sums over the pairs (i, k) that list j.  The kernels above add them with one float atomic per
// (pair, channel): 42 M atomics per pass at (40960, 16, 64), executed on the memory side of the fabric, run-to-run different in the last bits.
// Over the transposed neighbour table (cbl_neighbor_transpose: for every target row the ascending list of its pairs) they are gathers:
//   x_v : grad_xv[j, c] = sum over j's pairs p = (i, k) of a[p, c % G] * grad_out[i, c]                       — nothing to recompute
//   x_k : grad_xk[j, c] = sum over j's pairs of dw[p, c], the BatchNorm-backward value the apply pass forms per pair: recomputed here from
//         x_q[i], p1[p], grad_w2[p, :] and x_k[j] (the target's own row, read once) with the apply pass's arithmetic, operation for operation
// One group of C lanes (lane = channel) owns a target and walks its list; plain stores, no zero fill, deterministic.
template <int C, int G>
__global__ __launch_bounds__(AT_BLOCK) void attn_agg_gxv_csr_kernel(unsigned n, CblFastDiv dvK, const float* __restrict__ a, const float* __restrict__ go,
                                                                    const int* __restrict__ order, const int* __restrict__ inv_start, const int* __restrict__ inv_src,
                                                                    float* __restrict__ gxv)
{
    constexpr int GPB = AT_BLOCK / C;
    const int c = threadIdx.x % C, grp = threadIdx.x / C;
    const unsigned ntrips = (n + GPB - 1) / GPB;
    for (unsigned v = blockIdx.x; v < 8 * cbl_xcd_per(ntrips); v += gridDim.x) {
        const unsigned tr = cbl_xcd_slot(v, ntrips) * GPB + grp;
        if (tr >= n) continue;
        const int j = order ? order[tr] : (int)tr;
        const int e0 = inv_start[tr], e1 = inv_start[tr + 1];
        float acc = 0.f;
        int e = e0;
        for (; e + AT_U <= e1; e += AT_U) {
            int p[AT_U]; float av[AT_U], gv[AT_U];
#pragma unroll
            for (int u = 0; u < AT_U; u++) p[u] = inv_src[e + u];
#pragma unroll
            for (int u = 0; u < AT_U; u++) { av[u] = a[(size_t)p[u] * G + (c % G)]; gv[u] = go[(size_t)cbl_fastdiv((unsigned)p[u], dvK) * C + c]; }
#pragma unroll
            for (int u = 0; u < AT_U; u++) acc += gv[u] * av[u];
        }
        for (; e < e1; e++) { const int p = inv_src[e]; acc += go[(size_t)cbl_fastdiv((unsigned)p, dvK) * C + c] * a[(size_t)p * G + (c % G)]; }
        gxv[(size_t)j * C + c] = acc;
    }
}

template <int C, int G>
__global__ __launch_bounds__(AT_BLOCK) void attn_w2_gxk_csr_kernel(unsigned n, int K, CblFastDiv dvK, const float* __restrict__ xq, const float* __restrict__ xk,
                                                                   const float* __restrict__ p1, const float* __restrict__ W3C, const float* __restrict__ b3C,
                                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   const float* __restrict__ Wa, const float* __restrict__ gw2, const float* __restrict__ sums,
                                                                   const int* __restrict__ order, const int* __restrict__ inv_start, const int* __restrict__ inv_src,
                                                                   float* __restrict__ gxk)
{
    constexpr int GPB = AT_BLOCK / C;
    const int c = threadIdx.x % C, grp = threadIdx.x / C;
    PairParams q; q.w0 = W3C[3 * c]; q.w1 = W3C[3 * c + 1]; q.w2 = W3C[3 * c + 2]; q.b = b3C[c];
    q.mean = mean[c]; q.invstd = invstd[c]; q.gamma = gamma ? gamma[c] : 1.f; q.beta = beta ? beta[c] : 0.f;
    float wa[G];
#pragma unroll
    for (int g = 0; g < G; g++) wa[g] = Wa[g * C + c];
    const float inv_rows = 1.0f / ((float)n * (float)K);
    const float c0 = sums[c] * inv_rows, c1 = sums[C + c] * inv_rows;
    const unsigned ntrips = (n + GPB - 1) / GPB;
    for (unsigned v = blockIdx.x; v < 8 * cbl_xcd_per(ntrips); v += gridDim.x) {
        const unsigned tr = cbl_xcd_slot(v, ntrips) * GPB + grp;
        if (tr >= n) continue;
        const int j = order ? order[tr] : (int)tr;
        const int e0 = inv_start[tr], e1 = inv_start[tr + 1];
        const float xkj = xk[(size_t)j * C + c];
        float acc = 0.f;
        for (int eb = e0; eb < e1; eb += AT_U) {
            int p[AT_U]; float xqi[AT_U], a0[AT_U], a1[AT_U], a2[AT_U], gd[AT_U][G];
#pragma unroll
            for (int u = 0; u < AT_U; u++) p[u] = inv_src[min(eb + u, e1 - 1)];
#pragma unroll
            for (int u = 0; u < AT_U; u++) {
                xqi[u] = xq[(size_t)cbl_fastdiv((unsigned)p[u], dvK) * C + c];
                a0[u] = p1[3 * (size_t)p[u]]; a1[u] = p1[3 * (size_t)p[u] + 1]; a2[u] = p1[3 * (size_t)p[u] + 2];
#pragma unroll
                for (int g = 0; g < G; g++) gd[u][g] = gw2[(size_t)p[u] * G + g];
            }
#pragma unroll
            for (int u = 0; u < AT_U; u++) {
                if (eb + u < e1) {
                    const float w = pe_of(q, a0[u], a1[u], a2[u]) - (xqi[u] - xkj);
                    const float xh = (w - q.mean) * q.invstd;
                    const float y = xh * q.gamma + q.beta;
                    float dw1 = 0.f;
#pragma unroll
                    for (int g = 0; g < G; g++) dw1 += wa[g] * gd[u][g];
                    const float dy = y > 0.f ? dw1 : 0.f;
                    acc += q.gamma * q.invstd * ((dy - c0) - xh * c1);             // BatchNorm backward, train mode: the apply pass's dw
                }
            }
        }
        gxk[(size_t)j * C + c] = acc;
    }
}

// ----------------------------------------------------------------------------------------------------------- wide stages
// C = 128 / 256 / 512 (the three coarse stages: 2560 / 640 / 160 points per 40960-point scene, share_planes = 8 -> G = C / 8).  Their
// tensors are small; what they cost is LAUNCHES: ~75 per layer on the separate kernels, 16 of the network's 23 layers.  The same
// passes as above with ONE point per workgroup of C lanes (lane = channel, 2 / 4 / 8 waves): per-channel work is unchanged, the sums over
// a point's channels go wave (DPP) -> LDS -> workgroup.
// The contractions over a point's channels go to the matrix cores: the C lanes leave the values of 16 pairs in LDS as V (16 x C, row stride
// C + 4 so that the A-operand reads of v_mfma_f32_16x16x4_f32 hit every bank twice, the minimum for 64 lanes), each wave multiplies a slice of
// V's columns by its register-resident slice of the weight matrix (B operand), slices are summed through LDS.
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int WT = 16;                                           // pairs per tile (the MFMA's M)
template <int C> constexpr int wide_stride() { return C + 4; }

template <int C, int STEPS>
__device__ __forceinline__ f32x4 wide_mfma(const float* V, int step0, const float (&bw)[STEPS])
{
    const int lane = threadIdx.x & 63;
    const float* a = V + (lane & 15) * wide_stride<C>() + (lane >> 4) + 4 * step0;      // A[i][kk]: lane = i + 16 kk
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < STEPS; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * s], bw[s], acc, 0, 0, 0);
    return acc;                                                                          // D[4 (lane / 16) + r][lane % 16] in acc[r]
}

// (16 x C) . W3C (C x 3): the gradient of p1 = the pair's 3-vector, for 16 pairs at once.  All C/64 waves take C/(4 NW) = 16 k-steps each.
template <int C>
struct WideTimes3 {
    static constexpr int NW = C / 64, STEPS = 16;
    float bw[STEPS];
    __device__ __forceinline__ void load(const float* __restrict__ W3C)
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, kk = lane >> 4;
#pragma unroll
        for (int s = 0; s < STEPS; s++) bw[s] = j < 3 ? W3C[(size_t)(4 * (wave * STEPS + s) + kk) * 3 + j] : 0.f;   // B[kk][j]: lane = j + 16 kk
    }
    // P: NW x 16 x 4 floats of LDS; call between two barriers
    __device__ __forceinline__ void partial(const float* V, float* P) const
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const f32x4 acc = wide_mfma<C, STEPS>(V, wave * STEPS, bw);
        if ((lane & 15) < 3) {
#pragma unroll
            for (int r = 0; r < 4; r++) P[(wave * WT + 4 * (lane >> 4) + r) * 4 + (lane & 15)] = acc[r];
        }
    }
    // after the barrier: thread e < 48 owns gp1[u = e / 3][a = e % 3]
    __device__ __forceinline__ float total(const float* P, int e) const
    {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w++) t += P[(w * WT + e / 3) * 4 + e % 3];
        return t;
    }
};

template <int C>
__global__ __launch_bounds__(C) void attn_w2_stats_wide_kernel(int n, int K, const float* __restrict__ xq, const float* __restrict__ xk,
                                                               const int* __restrict__ idx, const float* __restrict__ p1,
                                                               const float* __restrict__ W3C, const float* __restrict__ b3C, float* __restrict__ partial)
{
    const int c = threadIdx.x;
    PairParams q; q.w0 = W3C[3 * c]; q.w1 = W3C[3 * c + 1]; q.w2 = W3C[3 * c + 2]; q.b = b3C[c];
    float s0 = 0.f, s1 = 0.f;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const float xqi = xq[(size_t)i * C + c];
        for (int k0 = 0; k0 < K; k0 += AT_U) {
            PairBatch pb; load_pairs<C>(pb, i, k0, K, c, idx, p1, xk);
#pragma unroll
            for (int u = 0; u < AT_U; u++)
                if (k0 + u < K) { const float w = pe_of(q, pb.a0[u], pb.a1[u], pb.a2[u]) - (xqi - pb.xr[u]); s0 += w; s1 += w * w; }
        }
    }
    partial[((size_t)blockIdx.x * 2) * C + c] = s0;
    partial[((size_t)blockIdx.x * 2 + 1) * C + c] = s1;
}

template <int C, int G>
__global__ __launch_bounds__(C) void attn_w2_forward_wide_kernel(int n, int K, const float* __restrict__ xq, const float* __restrict__ xk,
                                                                 const int* __restrict__ idx, const float* __restrict__ p1,
                                                                 const float* __restrict__ W3C, const float* __restrict__ b3C,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const float* __restrict__ Wa, const float* __restrict__ ba, float* __restrict__ w2)
{
    // w2 (16 pairs x G) = w1 (16 x C) . Wa^T (C x G): G/16 column tiles x 2 halves of the k-range = C/64 waves, C/8 MFMA steps each
    constexpr int NT = G / 16, STEPS = C / 8;
    __shared__ float V[WT * wide_stride<C>()];
    __shared__ float P[2 * WT * G];
    const int c = threadIdx.x, lane = c & 63, wave = c >> 6;
    const int tile = wave % NT, half = wave / NT;
    PairParams q; q.w0 = W3C[3 * c]; q.w1 = W3C[3 * c + 1]; q.w2 = W3C[3 * c + 2]; q.b = b3C[c];
    q.mean = mean[c]; q.invstd = invstd[c]; q.gamma = gamma ? gamma[c] : 1.f; q.beta = beta ? beta[c] : 0.f;
    float bw[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; s++) bw[s] = Wa[(size_t)(16 * tile + (lane & 15)) * C + 4 * (half * STEPS + s) + (lane >> 4)];
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const float xqi = xq[(size_t)i * C + c];
        for (int k0 = 0; k0 < K; k0 += WT) {
            float xr[WT];
#pragma unroll
            for (int u = 0; u < WT; u++) xr[u] = xk[(size_t)idx[(size_t)i * K + min(k0 + u, K - 1)] * C + c];
#pragma unroll
            for (int u = 0; u < WT; u++) {
                const size_t r = (size_t)i * K + min(k0 + u, K - 1);
                const float w = pe_of(q, p1[3 * r], p1[3 * r + 1], p1[3 * r + 2]) - (xqi - xr[u]);
                const float y = (w - q.mean) * q.invstd * q.gamma + q.beta;
                V[u * wide_stride<C>() + c] = y > 0.f ? y : 0.f;
            }
            __syncthreads();
            const f32x4 acc = wide_mfma<C, STEPS>(V, half * STEPS, bw);
#pragma unroll
            for (int r = 0; r < 4; r++) P[(half * WT + 4 * (lane >> 4) + r) * G + 16 * tile + (lane & 15)] = acc[r];
            __syncthreads();
            for (int e = c; e < WT * G; e += C) {
                const int u = e / G, g = e % G;
                if (k0 + u < K) w2[((size_t)i * K + k0) * G + e] = P[e] + P[WT * G + e] + ba[g];
            }
        }
    }
}

template <int C, int G>
__global__ __launch_bounds__(C) void attn_w2_bwd_reduce_wide_kernel(int n, int K, const float* __restrict__ xq, const float* __restrict__ xk,
                                                                    const int* __restrict__ idx, const float* __restrict__ p1,
                                                                    const float* __restrict__ W3C, const float* __restrict__ b3C,
                                                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    const float* __restrict__ Wa, const float* __restrict__ gw2, float* __restrict__ partial)
{
    const int c = threadIdx.x;
    PairParams q; q.w0 = W3C[3 * c]; q.w1 = W3C[3 * c + 1]; q.w2 = W3C[3 * c + 2]; q.b = b3C[c];
    q.mean = mean[c]; q.invstd = invstd[c]; q.gamma = gamma ? gamma[c] : 1.f; q.beta = beta ? beta[c] : 0.f;
    float wa[G], dwa[G];
#pragma unroll
    for (int g = 0; g < G; g++) { wa[g] = Wa[g * C + c]; dwa[g] = 0.f; }
    float s0 = 0.f, s1 = 0.f, dba = 0.f;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const float xqi = xq[(size_t)i * C + c];
        for (int k = 0; k < K; k++) {
            const size_t r = (size_t)i * K + k;
            const int j = idx[r];
            const float w = pe_of(q, p1[3 * r], p1[3 * r + 1], p1[3 * r + 2]) - (xqi - xk[(size_t)j * C + c]);
            const float xh = (w - q.mean) * q.invstd;
            const float y = xh * q.gamma + q.beta;
            const float w1 = y > 0.f ? y : 0.f;
            float dw1 = 0.f;
#pragma unroll
            for (int g = 0; g < G; g++) { const float gd = gw2[r * G + g]; dw1 += wa[g] * gd; dwa[g] += gd * w1; dba += (c == g) ? gd : 0.f; }
            const float dy = y > 0.f ? dw1 : 0.f;
            s0 += dy; s1 += dy * xh;
        }
    }
    float* mine = partial + (size_t)blockIdx.x * (2 * C + G * C + G);
    mine[c] = s0; mine[C + c] = s1;
#pragma unroll
    for (int g = 0; g < G; g++) mine[2 * C + g * C + c] = dwa[g];
    if (c < G) mine[2 * C + G * C + c] = dba;
}

template <int C, int G>
__global__ __launch_bounds__(C) void attn_w2_bwd_apply_wide_kernel(int n, int K, const float* __restrict__ xq, const float* __restrict__ xk,
                                                                   const int* __restrict__ idx, const float* __restrict__ p1,
                                                                   const float* __restrict__ W3C, const float* __restrict__ b3C,
                                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   const float* __restrict__ Wa, const float* __restrict__ gw2, const float* __restrict__ sums,
                                                                   float* __restrict__ gxq, float* __restrict__ gxk, float* __restrict__ gp1,
                                                                   float* __restrict__ partial)
{
    __shared__ float V[WT * wide_stride<C>()];
    __shared__ float P[(C / 64) * WT * 4];
    const int c = threadIdx.x;
    PairParams q; q.w0 = W3C[3 * c]; q.w1 = W3C[3 * c + 1]; q.w2 = W3C[3 * c + 2]; q.b = b3C[c];
    q.mean = mean[c]; q.invstd = invstd[c]; q.gamma = gamma ? gamma[c] : 1.f; q.beta = beta ? beta[c] : 0.f;
    float wa[G];
#pragma unroll
    for (int g = 0; g < G; g++) wa[g] = Wa[g * C + c];
    WideTimes3<C> t3; t3.load(W3C);
    const float inv_rows = 1.0f / ((float)n * (float)K);
    const float c0 = sums[c] * inv_rows, c1 = sums[C + c] * inv_rows;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, db = 0.f;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const float xqi = xq[(size_t)i * C + c];
        float gq = 0.f;
        for (int k0 = 0; k0 < K; k0 += WT) {
            int jj[WT]; float xr[WT];
#pragma unroll
            for (int u = 0; u < WT; u++) { jj[u] = idx[(size_t)i * K + min(k0 + u, K - 1)]; xr[u] = xk[(size_t)jj[u] * C + c]; }
#pragma unroll
            for (int u = 0; u < WT; u++) {
                const size_t r = (size_t)i * K + min(k0 + u, K - 1);
                const float a0 = p1[3 * r], a1 = p1[3 * r + 1], a2 = p1[3 * r + 2];
                const float w = pe_of(q, a0, a1, a2) - (xqi - xr[u]);
                const float xh = (w - q.mean) * q.invstd;
                const float y = xh * q.gamma + q.beta;
                float dw1 = 0.f;
#pragma unroll
                for (int g = 0; g < G; g++) dw1 += wa[g] * gw2[r * G + g];
                const float dy = y > 0.f ? dw1 : 0.f;
                float dw = q.gamma * q.invstd * ((dy - c0) - xh * c1);       // BatchNorm backward, train mode
                if (k0 + u >= K) dw = 0.f;
                else unsafeAtomicAdd(gxk + (size_t)jj[u] * C + c, dw);        // w = p_r - x_q + x_k[j]
                gq -= dw;
                d0 += dw * a0; d1 += dw * a1; d2 += dw * a2; db += dw;
                V[u * wide_stride<C>() + c] = dw;
            }
            __syncthreads();
            t3.partial(V, P);
            __syncthreads();
            if (c < 3 * WT && k0 + c / 3 < K) gp1[3 * ((size_t)i * K + k0) + c] = t3.total(P, c);
        }
        gxq[(size_t)i * C + c] = gq;
    }
    float* mine = partial + (size_t)blockIdx.x * (4 * C);
    mine[3 * c] = d0; mine[3 * c + 1] = d1; mine[3 * c + 2] = d2; mine[3 * C + c] = db;
}

template <int C, int G>
__global__ __launch_bounds__(C) void attn_agg_forward_wide_kernel(int n, int K, const float* __restrict__ xv, const int* __restrict__ idx,
                                                                  const float* __restrict__ p1, const float* __restrict__ W3C, const float* __restrict__ b3C,
                                                                  const float* __restrict__ a, float* __restrict__ a_out, float* __restrict__ out)
{
    const int c = threadIdx.x;
    PairParams q; q.w0 = W3C[3 * c]; q.w1 = W3C[3 * c + 1]; q.w2 = W3C[3 * c + 2]; q.b = b3C[c];
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        SoftmaxCol sc = {0.f, 1.f};
        if (a_out) sc = softmax_col(a, i, K, G, c % G);             // (as in the narrow kernel)
        float acc = 0.f;
        for (int k0 = 0; k0 < K; k0 += AT_U) {
            PairBatch pb; load_pairs<C>(pb, i, k0, K, c, idx, p1, xv);
            float av[AT_U];
#pragma unroll
            for (int u = 0; u < AT_U; u++) {
                av[u] = a[((size_t)i * K + min(k0 + u, K - 1)) * G + (c % G)];
                if (a_out) { av[u] = softmax_w(sc, av[u]); if (c < G && k0 + u < K) a_out[((size_t)i * K + k0 + u) * G + c] = av[u]; }
            }
#pragma unroll
            for (int u = 0; u < AT_U; u++)
                if (k0 + u < K) acc += (pb.xr[u] + pe_of(q, pb.a0[u], pb.a1[u], pb.a2[u])) * av[u];
        }
        out[(size_t)i * C + c] = acc;
    }
}

template <int C, int G>
__global__ __launch_bounds__(C) void attn_agg_backward_wide_kernel(int n, int K, const float* __restrict__ xv, const int* __restrict__ idx,
                                                                   const float* __restrict__ p1, const float* __restrict__ W3C, const float* __restrict__ b3C,
                                                                   const float* __restrict__ a, const float* __restrict__ go,
                                                                   float* __restrict__ gxv, float* __restrict__ gp1, float* __restrict__ ga, float* __restrict__ partial,
                                                                   int softmax)
{
    __shared__ float V[WT * wide_stride<C>()];
    __shared__ float DA[WT * C];
    __shared__ float P[(C / 64) * WT * 4];
    const int c = threadIdx.x;
    PairParams q; q.w0 = W3C[3 * c]; q.w1 = W3C[3 * c + 1]; q.w2 = W3C[3 * c + 2]; q.b = b3C[c];
    WideTimes3<C> t3; t3.load(W3C);
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, db = 0.f;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const float g = go[(size_t)i * C + c];
        float dotp = 0.f;                                            // softmax: this thread's share of sum_k a[k, g] da[k, g], g = c % G
        for (int k0 = 0; k0 < K; k0 += WT) {
            int jj[WT]; float xr[WT];
#pragma unroll
            for (int u = 0; u < WT; u++) { jj[u] = idx[(size_t)i * K + min(k0 + u, K - 1)]; xr[u] = xv[(size_t)jj[u] * C + c]; }
#pragma unroll
            for (int u = 0; u < WT; u++) {
                const size_t r = (size_t)i * K + min(k0 + u, K - 1);
                const float a0 = p1[3 * r], a1 = p1[3 * r + 1], a2 = p1[3 * r + 2];
                float dpe = g * a[r * G + (c % G)];                          // d out / d (x_v[j] + p_r)
                if (k0 + u >= K) dpe = 0.f;
                else unsafeAtomicAdd(gxv + (size_t)jj[u] * C + c, dpe);
                d0 += dpe * a0; d1 += dpe * a1; d2 += dpe * a2; db += dpe;
                V[u * wide_stride<C>() + c] = dpe;
                DA[u * C + c] = g * (xr[u] + pe_of(q, a0, a1, a2));          // grad_a[i,k,g] = sum over the channels with c % G == g
            }
            __syncthreads();
            t3.partial(V, P);
            for (int e = c; e < WT * G; e += C) {
                const int u = e / G, gg = e % G;
                float da = 0.f;
#pragma unroll
                for (int st = 0; st < C / G; st++) da += DA[u * C + gg + G * st];
                if (k0 + u < K) { ga[((size_t)i * K + k0) * G + e] = da; dotp += a[((size_t)i * K + k0) * G + e] * da; }
            }
            __syncthreads();
            if (c < 3 * WT && k0 + c / 3 < K) gp1[3 * ((size_t)i * K + k0) + c] = t3.total(P, c);
        }
        if (softmax) {                                               // gradient of the logits: a (da - sum over the point's pairs of a da), column by column
            DA[c] = dotp;                                            // (every thread's entries e = c, c + C, .. share the column c % G: G divides C)
            __syncthreads();
            float dot = 0.f;
#pragma unroll
            for (int st = 0; st < C / G; st++) dot += DA[(c % G) + G * st];
            for (int k0 = 0; k0 < K; k0 += WT)
                for (int e = c; e < WT * G; e += C)
                    if (k0 + e / G < K) { const size_t at = ((size_t)i * K + k0) * G + e; ga[at] = a[at] * (ga[at] - dot); }   // written by this thread above
            __syncthreads();
        }
    }
    float* mine = partial + (size_t)blockIdx.x * (4 * C);
    mine[3 * c] = d0; mine[3 * c + 1] = d1; mine[3 * c + 2] = d2; mine[3 * C + c] = db;
}

// partial rows per pass: one per workgroup; enough workgroups to fill 256 CUs, few enough that the (2C + GC + G)-float rows stay ~30 MB
// (C = 128: 2048 since round 5 — with several scenes per step the stage has 10^4 points and more, and the passes are latency chains per point: twice the workgroups
//  in flight took the layer from 1400 to 1214 us at 20480 points, 729 to 654 at 10240; four times measured no better; one scene has 2560 points = 2560 workgroups either way)
constexpr int at_wide_blocks(int C) { return C <= 128 ? 2048 : (C <= 256 ? 768 : 256); }

int at_check(int n, int K, int C, int G)
{
    if (n < 0 || K <= 0) return CBL_ERR_BAD_ARG;
    if (!((C == 32 && G == 4) || (C == 64 && G == 8) || (C == 128 && G == 16) || (C == 256 && G == 32) || (C == 512 && G == 64))) return CBL_ERR_UNSUPPORTED;
    return CBL_OK;
}
inline int at_blocks(int n, int C)
{
    if (C > 64) return n < 1 ? 1 : (n > at_wide_blocks(C) ? at_wide_blocks(C) : n);                     // wide stages: one point per workgroup and trip
    const int gpb = AT_BLOCK / C; const long long b = ((long long)n + gpb - 1) / gpb; return (int)(b < 1 ? 1 : (b > AT_MAX_BLOCKS ? AT_MAX_BLOCKS : b));
}

}  // namespace

// floats of scratch: partials of the widest pass (Q1) + the summed values
CBL_EXPORT size_t cbl_attn_workspace_bytes(int C, int G)
{
    const size_t per_block = (size_t)2 * C + (size_t)G * C + G;
    return sizeof(float) * ((C > 64 ? at_wide_blocks(C) : AT_MAX_BLOCKS) * per_block + per_block) + 256;
}

#define AT_DISPATCH(KERNEL, ...)                                                                                              \
    do {                                                                                                                          \
        if (C == 32)       hipLaunchKernelGGL((KERNEL##_kernel<32, 4>), dim3(nb), dim3(AT_BLOCK), 0, st, __VA_ARGS__);            \
        else if (C == 64)  hipLaunchKernelGGL((KERNEL##_kernel<64, 8>), dim3(nb), dim3(AT_BLOCK), 0, st, __VA_ARGS__);            \
        else if (C == 128) hipLaunchKernelGGL((KERNEL##_wide_kernel<128, 16>), dim3(nb), dim3(128), 0, st, __VA_ARGS__);          \
        else if (C == 256) hipLaunchKernelGGL((KERNEL##_wide_kernel<256, 32>), dim3(nb), dim3(256), 0, st, __VA_ARGS__);          \
        else               hipLaunchKernelGGL((KERNEL##_wide_kernel<512, 64>), dim3(nb), dim3(512), 0, st, __VA_ARGS__);          \
    } while (0)

CBL_EXPORT int cbl_attn_w2_forward(int n, int K, int C, int G, const float* x_q, const float* x_k, const int* idx, const float* p1,
                                   const float* W3C, const float* b3C, const float* bn_weight, const float* bn_bias, float eps, float momentum,
                                   float* running_mean, float* running_var, long long* num_batches_tracked, int training,
                                   const float* Wa, const float* ba, float* save_mean, float* save_invstd, float* w2,
                                   void* workspace, size_t workspace_bytes, void* stream)
{
    const int rc = at_check(n, K, C, G);
    if (rc) return rc;
    if (n == 0) return CBL_OK;
    if (!x_q || !x_k || !idx || !p1 || !W3C || !b3C || !Wa || !ba || !save_mean || !save_invstd || !w2 || !workspace) return CBL_ERR_BAD_ARG;
    if (workspace_bytes < cbl_attn_workspace_bytes(C, G)) return CBL_ERR_WORKSPACE;
    hipStream_t st = cbl_stream(stream);
    const int nb = at_blocks(n, C);
    float* partial = reinterpret_cast<float*>(workspace);
    if (training) {                                                   // eval mode: the caller put the running statistics into save_mean / save_invstd
        if (C == 32)       hipLaunchKernelGGL(attn_w2_stats_kernel<32>, dim3(nb), dim3(AT_BLOCK), 0, st, n, K, x_q, x_k, idx, p1, W3C, b3C, partial);
        else if (C == 64)  hipLaunchKernelGGL(attn_w2_stats_kernel<64>, dim3(nb), dim3(AT_BLOCK), 0, st, n, K, x_q, x_k, idx, p1, W3C, b3C, partial);
        else if (C == 128) hipLaunchKernelGGL(attn_w2_stats_wide_kernel<128>, dim3(nb), dim3(128), 0, st, n, K, x_q, x_k, idx, p1, W3C, b3C, partial);
        else if (C == 256) hipLaunchKernelGGL(attn_w2_stats_wide_kernel<256>, dim3(nb), dim3(256), 0, st, n, K, x_q, x_k, idx, p1, W3C, b3C, partial);
        else               hipLaunchKernelGGL(attn_w2_stats_wide_kernel<512>, dim3(nb), dim3(512), 0, st, n, K, x_q, x_k, idx, p1, W3C, b3C, partial);
        hipLaunchKernelGGL(attn_bn_finalize_kernel, dim3(cbl_div_up(C, 16)), dim3(256), 0, st, (long long)n * K, C, nb, partial, eps, momentum,
                           running_mean, running_var, num_batches_tracked, save_mean, save_invstd);
    }
    if ((C == 32 || C == 64) && cbl_host_aligned16(x_q) && cbl_host_aligned16(x_k)) {
        const long long tiles = (16 % K) == 0 ? ((long long)n * K + 15) / 16 : (long long)n * ((K + 15) / 16);
        const unsigned g = (unsigned)min((tiles + 3) / 4, (long long)8192);
        if (C == 32) hipLaunchKernelGGL(attn_w2_forward_mfma_kernel<32>, dim3(g), dim3(AT_BLOCK), 0, st, n, K, x_q, x_k, idx, p1, W3C, b3C, save_mean, save_invstd, bn_weight, bn_bias, Wa, ba, w2);
        else         hipLaunchKernelGGL(attn_w2_forward_mfma_kernel<64>, dim3(g), dim3(AT_BLOCK), 0, st, n, K, x_q, x_k, idx, p1, W3C, b3C, save_mean, save_invstd, bn_weight, bn_bias, Wa, ba, w2);
        return cbl_status();
    }
    AT_DISPATCH(attn_w2_forward, n, K, x_q, x_k, idx, p1, W3C, b3C, save_mean, save_invstd, bn_weight, bn_bias, Wa, ba, w2);
    return cbl_status();
}

CBL_EXPORT int cbl_attn_w2_backward(int n, int K, int C, int G, const float* x_q, const float* x_k, const int* idx, const float* p1,
                                    const float* W3C, const float* b3C, const float* bn_weight, const float* bn_bias,
                                    const float* save_mean, const float* save_invstd, const float* Wa, const float* grad_w2,
                                    float* grad_xq, float* grad_xk, float* grad_p1, float* grad_W3C, float* grad_b3C,
                                    float* grad_bn_weight, float* grad_bn_bias, float* grad_Wa, float* grad_ba,
                                    void* workspace, size_t workspace_bytes, void* stream)
{
    const int rc = at_check(n, K, C, G);
    if (rc) return rc;
    if (n == 0) return CBL_OK;
    if (!x_q || !x_k || !idx || !p1 || !W3C || !b3C || !save_mean || !save_invstd || !Wa || !grad_w2 || !grad_xq || !grad_xk || !grad_p1 || !grad_W3C ||
        !grad_b3C || !grad_bn_weight || !grad_bn_bias || !grad_Wa || !grad_ba || !workspace) return CBL_ERR_BAD_ARG;
    if (workspace_bytes < cbl_attn_workspace_bytes(C, G)) return CBL_ERR_WORKSPACE;
    hipStream_t st = cbl_stream(stream);
    const int nb = at_blocks(n, C);
    const int nv1 = 2 * C + G * C + G, nv2 = 4 * C;
    float* partial = reinterpret_cast<float*>(workspace);
    float* sums = partial + (size_t)(C > 64 ? at_wide_blocks(C) : AT_MAX_BLOCKS) * nv1;
    AT_DISPATCH(attn_w2_bwd_reduce, n, K, x_q, x_k, idx, p1, W3C, b3C, save_mean, save_invstd, bn_weight, bn_bias, Wa, grad_w2, partial);
    // Q1's sums = [grad_beta | grad_gamma | grad_Wa | grad_ba]: written to the four outputs and kept in `sums` for Q2 (BatchNorm's two means)
    SumSegments s1; s1.dst[0] = grad_bn_bias; s1.dst[1] = grad_bn_weight; s1.dst[2] = grad_Wa; s1.dst[3] = grad_ba;
    s1.begin[0] = 0; s1.begin[1] = C; s1.begin[2] = 2 * C; s1.begin[3] = 2 * C + G * C; s1.begin[4] = nv1;
    hipLaunchKernelGGL(attn_sum_partials_kernel, dim3(cbl_div_up(nv1, 16)), dim3(256), 0, st, nv1, nb, partial, s1, sums);
    AT_DISPATCH(attn_w2_bwd_apply, n, K, x_q, x_k, idx, p1, W3C, b3C, save_mean, save_invstd, bn_weight, bn_bias, Wa, grad_w2, sums,
                grad_xq, grad_xk, grad_p1, partial);
    // partial rows are [dW3C (C x 3, row-major like the weight) | db3C]: summed straight into the two outputs
    SumSegments s2; s2.dst[0] = grad_W3C; s2.dst[1] = grad_b3C; s2.dst[2] = s2.dst[3] = nullptr;
    s2.begin[0] = 0; s2.begin[1] = 3 * C; s2.begin[2] = s2.begin[3] = s2.begin[4] = nv2;
    hipLaunchKernelGGL(attn_sum_partials_kernel, dim3(cbl_div_up(nv2, 16)), dim3(256), 0, st, nv2, nb, partial, s2, (float*)nullptr);
    return cbl_status();
}

static int attn_agg_forward_impl(int n, int K, int C, int G, const float* x_v, const int* idx, const float* p1, const float* W3C, const float* b3C,
                                 const float* a, float* a_out, float* out, void* stream)
{
    const int rc = at_check(n, K, C, G);
    if (rc) return rc;
    if (n == 0) return CBL_OK;
    if (!x_v || !idx || !p1 || !W3C || !b3C || !a || !out) return CBL_ERR_BAD_ARG;
    hipStream_t st = cbl_stream(stream);
    const int nb = C > 64 ? (n < 1 ? 1 : (n > 2048 ? 2048 : n)) : (int)cbl_grid_for((long long)n * C, AT_BLOCK, 4096);   // no partial rows here: any grid
    AT_DISPATCH(attn_agg_forward, n, K, x_v, idx, p1, W3C, b3C, a, a_out, out);
    return cbl_status();
}

CBL_EXPORT int cbl_attn_agg_forward(int n, int K, int C, int G, const float* x_v, const int* idx, const float* p1, const float* W3C, const float* b3C,
                                    const float* a, float* out, void* stream)
{
    return attn_agg_forward_impl(n, K, C, G, x_v, idx, p1, W3C, b3C, a, nullptr, out, stream);
}

// the same with the softmax over the K neighbours (blocks.py:41) inside: `logits` (n, K, G) in, the softmax weights out to `a` (n, K, G) for the
// backward pass (cbl_attn_agg_softmax_backward), `out` as cbl_attn_agg_forward would give for those weights
CBL_EXPORT int cbl_attn_agg_softmax_forward(int n, int K, int C, int G, const float* x_v, const int* idx, const float* p1, const float* W3C, const float* b3C,
                                            const float* logits, float* a, float* out, void* stream)
{
    if (!a) return CBL_ERR_BAD_ARG;
    return attn_agg_forward_impl(n, K, C, G, x_v, idx, p1, W3C, b3C, logits, a, out, stream);
}

static int attn_agg_backward_impl(int n, int K, int C, int G, const float* x_v, const int* idx, const float* p1, const float* W3C, const float* b3C,
                                  const float* a, const float* grad_out, float* grad_xv, float* grad_p1, float* grad_W3C, float* grad_b3C, float* grad_a,
                                  void* workspace, size_t workspace_bytes, int softmax, void* stream)
{
    const int rc = at_check(n, K, C, G);
    if (rc) return rc;
    if (n == 0) return CBL_OK;
    if (!x_v || !idx || !p1 || !W3C || !b3C || !a || !grad_out || !grad_xv || !grad_p1 || !grad_W3C || !grad_b3C || !grad_a || !workspace) return CBL_ERR_BAD_ARG;
    if (workspace_bytes < cbl_attn_workspace_bytes(C, G)) return CBL_ERR_WORKSPACE;
    hipStream_t st = cbl_stream(stream);
    const int nb = at_blocks(n, C);
    const int nv2 = 4 * C;
    float* partial = reinterpret_cast<float*>(workspace);
    AT_DISPATCH(attn_agg_backward, n, K, x_v, idx, p1, W3C, b3C, a, grad_out, grad_xv, grad_p1, grad_a, partial, softmax);
    SumSegments s2; s2.dst[0] = grad_W3C; s2.dst[1] = grad_b3C; s2.dst[2] = s2.dst[3] = nullptr;
    s2.begin[0] = 0; s2.begin[1] = 3 * C; s2.begin[2] = s2.begin[3] = s2.begin[4] = nv2;
    hipLaunchKernelGGL(attn_sum_partials_kernel, dim3(cbl_div_up(nv2, 16)), dim3(256), 0, st, nv2, nb, partial, s2, (float*)nullptr);
    return cbl_status();
}

// the same backward passes with the two scatters (grad_xk, grad_xv) as gathers over the transposed table of idx (C = 32 / 64; CBL_ERR_UNSUPPORTED
// otherwise: the wide stages are small and keep their atomics): grad_xk / grad_xv are WRITTEN (no pre-zeroing), no atomics anywhere, deterministic
CBL_EXPORT int cbl_attn_w2_backward_csr(int n, int K, int C, int G, const float* x_q, const float* x_k, const int* idx, const float* p1,
                                        const float* W3C, const float* b3C, const float* bn_weight, const float* bn_bias,
                                        const float* save_mean, const float* save_invstd, const float* Wa, const float* grad_w2,
                                        const int* order, const int* inv_start, const int* inv_src,
                                        float* grad_xq, float* grad_xk, float* grad_p1, float* grad_W3C, float* grad_b3C,
                                        float* grad_bn_weight, float* grad_bn_bias, float* grad_Wa, float* grad_ba,
                                        void* workspace, size_t workspace_bytes, void* stream)
{
    const int rc = at_check(n, K, C, G);
    if (rc) return rc;
    if (C > 64) return CBL_ERR_UNSUPPORTED;
    if (n == 0) return CBL_OK;
    if (!x_q || !x_k || !idx || !p1 || !W3C || !b3C || !save_mean || !save_invstd || !Wa || !grad_w2 || !grad_xq || !grad_xk || !grad_p1 || !grad_W3C ||
        !grad_b3C || !grad_bn_weight || !grad_bn_bias || !grad_Wa || !grad_ba || !workspace || !inv_start || !inv_src) return CBL_ERR_BAD_ARG;
    if (workspace_bytes < cbl_attn_workspace_bytes(C, G)) return CBL_ERR_WORKSPACE;
    hipStream_t st = cbl_stream(stream);
    const int nb = at_blocks(n, C);
    const int nv1 = 2 * C + G * C + G, nv2 = 4 * C;
    float* partial = reinterpret_cast<float*>(workspace);
    float* sums = partial + (size_t)AT_MAX_BLOCKS * nv1;
    AT_DISPATCH(attn_w2_bwd_reduce, n, K, x_q, x_k, idx, p1, W3C, b3C, save_mean, save_invstd, bn_weight, bn_bias, Wa, grad_w2, partial);
    SumSegments s1; s1.dst[0] = grad_bn_bias; s1.dst[1] = grad_bn_weight; s1.dst[2] = grad_Wa; s1.dst[3] = grad_ba;
    s1.begin[0] = 0; s1.begin[1] = C; s1.begin[2] = 2 * C; s1.begin[3] = 2 * C + G * C; s1.begin[4] = nv1;
    hipLaunchKernelGGL(attn_sum_partials_kernel, dim3(cbl_div_up(nv1, 16)), dim3(256), 0, st, nv1, nb, partial, s1, sums);
    AT_DISPATCH(attn_w2_bwd_apply, n, K, x_q, x_k, idx, p1, W3C, b3C, save_mean, save_invstd, bn_weight, bn_bias, Wa, grad_w2, sums,
                grad_xq, (float*)nullptr, grad_p1, partial);
    SumSegments s2; s2.dst[0] = grad_W3C; s2.dst[1] = grad_b3C; s2.dst[2] = s2.dst[3] = nullptr;
    s2.begin[0] = 0; s2.begin[1] = 3 * C; s2.begin[2] = s2.begin[3] = s2.begin[4] = nv2;
    hipLaunchKernelGGL(attn_sum_partials_kernel, dim3(cbl_div_up(nv2, 16)), dim3(256), 0, st, nv2, nb, partial, s2, (float*)nullptr);
    const CblFastDiv dv = cbl_fastdiv_make((unsigned)K);
    const unsigned g = cbl_round_up8((unsigned)cbl_grid_for((long long)n * C, AT_BLOCK, 8192));
    if (C == 32) hipLaunchKernelGGL((attn_w2_gxk_csr_kernel<32, 4>), dim3(g), dim3(AT_BLOCK), 0, st, (unsigned)n, K, dv, x_q, x_k, p1, W3C, b3C, save_mean, save_invstd, bn_weight, bn_bias, Wa, grad_w2, sums, order, inv_start, inv_src, grad_xk);
    else         hipLaunchKernelGGL((attn_w2_gxk_csr_kernel<64, 8>), dim3(g), dim3(AT_BLOCK), 0, st, (unsigned)n, K, dv, x_q, x_k, p1, W3C, b3C, save_mean, save_invstd, bn_weight, bn_bias, Wa, grad_w2, sums, order, inv_start, inv_src, grad_xk);
    return cbl_status();
}

CBL_EXPORT int cbl_attn_agg_backward_csr(int n, int K, int C, int G, const float* x_v, const int* idx, const float* p1, const float* W3C, const float* b3C,
                                         const float* a, const float* grad_out, const int* order, const int* inv_start, const int* inv_src,
                                         float* grad_xv, float* grad_p1, float* grad_W3C, float* grad_b3C, float* grad_a,
                                         void* workspace, size_t workspace_bytes, int softmax, void* stream)
{
    const int rc = at_check(n, K, C, G);
    if (rc) return rc;
    if (C > 64) return CBL_ERR_UNSUPPORTED;
    if (n == 0) return CBL_OK;
    if (!x_v || !idx || !p1 || !W3C || !b3C || !a || !grad_out || !grad_xv || !grad_p1 || !grad_W3C || !grad_b3C || !grad_a || !workspace || !inv_start || !inv_src)
        return CBL_ERR_BAD_ARG;
    if (workspace_bytes < cbl_attn_workspace_bytes(C, G)) return CBL_ERR_WORKSPACE;
    hipStream_t st = cbl_stream(stream);
    const int nb = at_blocks(n, C);
    const int nv2 = 4 * C;
    float* partial = reinterpret_cast<float*>(workspace);
    AT_DISPATCH(attn_agg_backward, n, K, x_v, idx, p1, W3C, b3C, a, grad_out, (float*)nullptr, grad_p1, grad_a, partial, softmax);
    SumSegments s2; s2.dst[0] = grad_W3C; s2.dst[1] = grad_b3C; s2.dst[2] = s2.dst[3] = nullptr;
    s2.begin[0] = 0; s2.begin[1] = 3 * C; s2.begin[2] = s2.begin[3] = s2.begin[4] = nv2;
    hipLaunchKernelGGL(attn_sum_partials_kernel, dim3(cbl_div_up(nv2, 16)), dim3(256), 0, st, nv2, nb, partial, s2, (float*)nullptr);
    // (with softmax the kernel above has replaced grad_a by the gradient of the logits; the gather needs the softmax WEIGHTS `a`, which are its input)
    const CblFastDiv dv = cbl_fastdiv_make((unsigned)K);
    const unsigned g = cbl_round_up8((unsigned)cbl_grid_for((long long)n * C, AT_BLOCK, 8192));
    if (C == 32) hipLaunchKernelGGL((attn_agg_gxv_csr_kernel<32, 4>), dim3(g), dim3(AT_BLOCK), 0, st, (unsigned)n, dv, a, grad_out, order, inv_start, inv_src, grad_xv);
    else         hipLaunchKernelGGL((attn_agg_gxv_csr_kernel<64, 8>), dim3(g), dim3(AT_BLOCK), 0, st, (unsigned)n, dv, a, grad_out, order, inv_start, inv_src, grad_xv);
    return cbl_status();
}

CBL_EXPORT int cbl_attn_agg_backward(int n, int K, int C, int G, const float* x_v, const int* idx, const float* p1, const float* W3C, const float* b3C,
                                     const float* a, const float* grad_out, float* grad_xv, float* grad_p1, float* grad_W3C, float* grad_b3C, float* grad_a,
                                     void* workspace, size_t workspace_bytes, void* stream)
{
    return attn_agg_backward_impl(n, K, C, G, x_v, idx, p1, W3C, b3C, a, grad_out, grad_xv, grad_p1, grad_W3C, grad_b3C, grad_a, workspace, workspace_bytes, 0, stream);
}

// backward of cbl_attn_agg_softmax_forward: `a` = the softmax weights it wrote, grad_logits (n, K, G) = a (da - sum over K of a da)
CBL_EXPORT int cbl_attn_agg_softmax_backward(int n, int K, int C, int G, const float* x_v, const int* idx, const float* p1, const float* W3C, const float* b3C,
                                             const float* a, const float* grad_out, float* grad_xv, float* grad_p1, float* grad_W3C, float* grad_b3C,
                                             float* grad_logits, void* workspace, size_t workspace_bytes, void* stream)
{
    return attn_agg_backward_impl(n, K, C, G, x_v, idx, p1, W3C, b3C, a, grad_out, grad_xv, grad_p1, grad_W3C, grad_b3C, grad_logits, workspace, workspace_bytes, 1,
                                  stream);
}
