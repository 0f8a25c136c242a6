// a4: PointTransformerLayer's vector attention (C = 32 / 64, K = 8 / 16: the two full-resolution stages), one pass structure for the whole layer.
// Reference: /root/reference/pytorch/model/blocks.py:31-44 (train-mode BatchNorms), with x_q / x_k / x_v = the three Linear layers' outputs (:33):
//     p_r = p_j - p_i ; p0 = Linear(3,3)(p_r) ; p1 = ReLU(BN_p(p0))                              (n,K,3)
//     pe  = Linear(3,C)(p1)                                                                       (n,K,C)   never stored
//     w   = x_k[j] - x_q[i] + pe ; w1 = ReLU(BN_c(w)) ; w2 = Linear(C,G)(w1), G = C/8             (n,K,G)
//     w3  = ReLU(BN_g(w2)) ; logits = Linear(G,G)(w3) ; a = softmax over K                        (n,K,G)
//     out = sum_k (x_v[j] + pe) * a[.., c % G]                                                    (n,C)
// The three train-mode BatchNorms are global synchronisation points; between them the layer is five passes over the (point, neighbour)
// pairs in which every C-wide pair value lives in registers and is RECOMPUTED where it is needed again:
//   forward   pchain (p_r, p0 + BN_p sums) | wstats (p1, BN_c sums) | w2 (+ BN_g sums) | softmax (narrow) | agg
//   backward  agg_bwd (d logits) | narrow_bwd (d w2 before BN_g, its sums) | reduce (d w2, BN_c sums, d Wa) | apply (d x_q, d p1, d W3C)
//             | pchain_bwd | target (d x_k, d x_v as gathers over the transposed neighbour table: no atomics)
// A TILE is 16 consecutive pairs of the flat (n K) list (K = 16: one point, K = 8: two points) = the 16 rows / columns of
// v_mfma_f32_16x16x4_f32; one wave owns a tile per trip.  With lo = lane % 16, hi = lane / 16 the kernels use two register layouts:
//   "pair-major"    lane (lo, hi) holds pair slot lo, channels 16 ct + 4 hi + v   (16-byte loads of a row; the contraction index of an MFMA step
//                   is hi, so sums over CHANNELS run on the matrix cores: w2 = Wa . w1)
//   "channel-major" lane (lo, hi) holds channel 16 ct + lo of pair slots 4 hi + v (64-byte row segments; this is the D layout of the
//                   instruction, and a D register fed back as the B operand contracts over PAIRS: the weight gradients d Wa, d W3C accumulate
//                   in MFMA accumulators over all tiles of a wave with no vector instruction)
// pe (a 4-term contraction with [p1, 1]) is ONE MFMA per 16 channels in either layout (operands swapped), its C input carrying x_k - x_q
// (or x_v), so the chain w = ((((x_k - x_q) + W0 p1_0) + W1 p1_1) + W2 p1_2) + b has the same bits in every pass (the ReLU masks agree);
// the target pass rebuilds it with fmaf in the same order.  Tiles are walked in the search's cell order, an XCD taking a contiguous eighth.
// BatchNorm sums: per-workgroup partial rows, reduced in double by the finalize kernels in a fixed order (deterministic, no atomics).
#include "cbl_common.h"
#include <pt_wave.h>
#include <cstdlib>

namespace {

#ifndef PT_BLOCK_THREADS
#define PT_BLOCK_THREADS 512
#endif
#ifndef PT_BWD_WAVES
#define PT_BWD_WAVES 2                 // waves per SIMD the two C-wide backward passes are compiled for (amdgpu_waves_per_eu)
#endif
constexpr int PT_BLOCK = PT_BLOCK_THREADS;                       // 8 waves.  Round 6, -DPT_BLOCK_THREADS=256 -DPT_BWD_WAVES=3 (three 4-wave workgroups per CU for the backward passes, constants
                                                                 // in LDS tables) against this at (40960, 16, 64): apply 56.1 -> 50.8 us, reduce 53.6 -> 57.8, softmax + aggregation 32.7 -> 38.3, statistics 22.0 -> 24.8
                                                                 // (every workgroup pays the in-consumer finalize): forward + backward 402 -> 421 us.  Kept at 512.
constexpr int PT_WPB = PT_BLOCK / 64;
constexpr int PT_MAX_ROWS = PT_BLOCK == 512 ? 512 : 1024;     // partial rows of a pass = its workgroups (all resident: two 512-lane or four 256-lane workgroups per CU)
constexpr int PT_ONE_PER_CU = 256 * (PT_BLOCK == 512 ? 1 : PT_BWD_WAVES);     // workgroups of a pass whose registers allow PT_BWD_WAVES waves per SIMD
constexpr int PT_NARROW_BLOCK = 256;
// pairs in flight per target and trip of the target pass: measured at (40960, 16, 64) 2: 51.4, 3: 52.0, 4: 59.1 us (196 registers: two waves per SIMD — the pass
// needs its waves more than deeper trips), at (40960, 8, 32) 2: 20.4, 3: 18.6, 4: 20.5 us
#ifndef PT_TARGET_PAIRS
#define PT_TARGET_PAIRS(C) ((C) == 32 ? 3 : 2)
#endif
// element indices inside the tile passes are 32-bit (one v_lshl_add_u64 per address instead of a sign extension, a 64-bit multiply and a 64-bit add: a third of the
// address arithmetic of a tile); pt_shape_ok bounds n K max(3, C / 8) and n C below 2^32
using pt_ix = unsigned;

// forward constants (floats) written by the finalize kernels, kept for the backward pass
constexpr int PT_CST_P = 0;                         // [4][4]  scale, shift, mean, invstd of BN_p (3 channels)
constexpr int PT_CST_C = 16;                        // [4][64] BN_c
constexpr int PT_CST_G = 272;                       // [4][8]  BN_g
constexpr int PT_FS_P = 304;                        // raw sums of the p chain: p0 [3], p0^2 [3], p_r [3], p0[a] p_r[b] [9]
constexpr int PT_FS_G = 336;                        // raw sums of w2 [8]
constexpr int PT_CST_FLOATS = 352;
// backward constants: d x = A1 d y + A2 x + A3 per channel (BatchNorm backward with the batch means folded in)
constexpr int PT_BC_G = 0;                          // [3][8]
constexpr int PT_BC_C = 24;                         // [3][64]
constexpr int PT_BC_FLOATS = 216;

__device__ __forceinline__ float pt_row_sum(float v)            // over the 16 lanes of a row, result in every lane
{
    v += pt_quad_xor1(v); v += pt_quad_xor2(v); v += pt_half_mirror(v); v += pt_row_mirror(v);
    return v;
}
__device__ __forceinline__ float pt_wave_sum(float v) { v = pt_row_sum(v); v += pt_xor16(v); v += pt_xor32(v); return v; }
template <int K> __device__ __forceinline__ float pt_group_sum(float v)      // over K = 8 / 16 consecutive lanes
{
    v += pt_quad_xor1(v); v += pt_quad_xor2(v); v += pt_half_mirror(v);
    if (K == 16) v += pt_row_mirror(v);
    return v;
}
template <int K> __device__ __forceinline__ float pt_group_max(float v)
{
    v = fmaxf(v, pt_quad_xor1(v)); v = fmaxf(v, pt_quad_xor2(v)); v = fmaxf(v, pt_half_mirror(v));
    if (K == 16) v = fmaxf(v, pt_row_mirror(v));
    return v;
}
// sum over the pair slots of one point held channel-major: the rows hi of the point (K = 16: all four, K = 8: the two of its half)
template <int K> __device__ __forceinline__ float pt_point_sum(float v)
{
    v += pt_xor16(v);
    if (K == 16) v += pt_xor32(v);
    return v;
}

// ---- tile geometry -------------------------------------------------------------------------------------------------------------------
// slot s of tile t is pair (point = order[t * (16 / K) + s / K], k = s % K).  A lane needs two slots: lo (pair-major operands) and 4 hi .. 4 hi + 3
// (channel-major values; K is a multiple of 4, so the four share a point).
struct PtS0 { int iA, iD, pA, pD; bool vA, vD, live; };
template <int K> __device__ __forceinline__ PtS0 pt_stage0(unsigned tile, unsigned ntiles, int n, const int* __restrict__ order, int lo, int hi)
{
    PtS0 r;
    r.live = tile < ntiles;
    const unsigned t = r.live ? tile : ntiles - 1;                   // a tile beyond the range: harmless addresses, nothing computed
    const unsigned ra = t * (16 / K) + (unsigned)(lo / K), rd = t * (16 / K) + (unsigned)((4 * hi) / K);
    r.vA = r.live && ra < (unsigned)n; r.vD = r.live && rd < (unsigned)n;
    const unsigned ca = ra < (unsigned)n ? ra : (unsigned)(n - 1), cd = rd < (unsigned)n ? rd : (unsigned)(n - 1);
    r.iA = order ? order[ca] : (int)ca;
    r.iD = (K == 16) ? r.iA : (order ? order[cd] : (int)cd);
    r.pA = r.iA * K + (lo % K);
    r.pD = r.iD * K + ((4 * hi) % K);
    return r;
}

// Software pipeline of a tile pass.  A tile's loads form a chain of three dependent levels — order[rank] -> the pair's neighbour id and narrow
// values -> the gathered rows — and a wave that walks its tiles one after the other pays the three latencies per tile (measured: 0.6 - 0.76 of
// all wave cycles parked on memory, 3 us per tile).  Here every level of a LATER tile is requested before the current tile is computed:
// level 0 three tiles ahead, level 1 two, level 2 (the rows) one.  Two tiles per trip of the loop, so that the two row buffers alternate
// without register copies (a copy of a just-requested value would wait for it); inside a phase the small levels are requested BEFORE the rows:
// memory returns in request order, so rotating the small values at the end of the trip waits for them only, the rows stay in flight.
// `pre` runs once, after the first tiles' loads are issued and before the first compute: a pass's in-consumer finalize (pt_fin_forward / pt_fin_backward) waits
// for its own round trip under the prefetch instead of in front of it.
struct PtNoPre { __device__ __forceinline__ void operator()() const {} };
template <class L0, class L1, class L2, class CP, class PRE = PtNoPre>
__device__ __forceinline__ void pt_pipeline(unsigned ntiles, L0 load0, L1 load1, L2 load2, CP compute, PRE pre = PRE())
{
    const unsigned ntrips = (ntiles + PT_WPB - 1) / PT_WPB, vend = 8u * cbl_xcd_per(ntrips), step = gridDim.x, wave = threadIdx.x >> 6;
    auto tile_of = [&](unsigned v) -> unsigned { return v < vend ? cbl_xcd_slot(v, ntrips) * PT_WPB + wave : 0xffffffffu; };
    unsigned v = blockIdx.x;
    auto a0 = load0(tile_of(v)); auto a1 = load0(tile_of(v + step)); auto a2 = load0(tile_of(v + 2 * step));
    auto b0 = load1(a0); auto b1 = load1(a1);
    auto c0 = load2(a0, b0);
    pre();
    for (; v < vend; v += 2 * step) {
        auto a3 = load0(tile_of(v + 3 * step));
        auto b2 = load1(a2);
        auto c1 = load2(a1, b1);
        if (a0.live) compute(a0, b0, c0);
        auto a4 = load0(tile_of(v + 4 * step));
        auto b3 = load1(a3);
        c0 = load2(a2, b2);
        if (a1.live) compute(a1, b1, c1);
        a0 = a2; a1 = a3; a2 = a4; b0 = b2; b1 = b3;
    }
}

// ---- LDS staging of a tile's rows --------------------------------------------------------------------------------------------------------
// The matrix instruction wants lane l to hold an element of pair slot l % 16: loading a gathered row in that layout makes every four adjacent
// lanes touch four different rows — the texture path then spends a cycle per LANE instead of per 64 bytes (measured: the first version of these
// passes sat at ~45 cycles per 16-byte gather instruction, 0.16 VALU utilisation, with every load already prefetched).  So the rows travel as
// rows: 16 lanes fetch the 16 consecutive 16-byte pieces of one row (four rows per instruction, whole 128-byte lines), the pieces wait in
// registers while the previous tile is computed, are written to a padded LDS tile of the wave, and each pass reads the tile back in the layout
// its arithmetic wants (pair-major: one ds_read_b128 per 16 channels; channel-major: ds_read_b32).  Rows 16.. hold the rows of the tile's own
// point(s): x_q[i] and / or d out[i].
constexpr int PT_ROWF = 68;                         // floats per staged row: 64 + 4 (rows land 4 banks apart: conflict-free 16-byte reads down a column)
constexpr int PT_TROWS = 21;                        // 16 neighbour rows, 2 x up to 2 point rows, the tile's p1 values [slot][3] (apply pass)
template <int NX> struct PtStaged { float4 k0, k1, k2, k3, x0, x1, pv; };     // named members: an array here ends up in scratch memory (measured)

// lane (lo, hi): piece lo of the rows of slots 4 hi .. 4 hi + 3 (ids j) and, in the first lane row of a point, of that point's rows in x0 / x1
template <int C, int K, int NX, bool NP = false>
__device__ __forceinline__ PtStaged<NX> pt_stage_rows(const float* __restrict__ rows, const int4& j, const float* __restrict__ xa, const float* __restrict__ xb,
                                                      int iD, int lo, int hi, const float* __restrict__ p1 = nullptr)
{
    PtStaged<NX> r;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const int piece = lo < C / 4 ? lo : 0;                           // C = 32: a row is 8 pieces; the upper lanes fetch piece 0 again (never read back)
    r.k0 = *reinterpret_cast<const float4*>(rows + (pt_ix)j.x * C + 4 * piece);
    r.k1 = *reinterpret_cast<const float4*>(rows + (pt_ix)j.y * C + 4 * piece);
    r.k2 = *reinterpret_cast<const float4*>(rows + (pt_ix)j.z * C + 4 * piece);
    r.k3 = *reinterpret_cast<const float4*>(rows + (pt_ix)j.w * C + 4 * piece);
    r.x0 = zero; r.x1 = zero;
    if (NX >= 1 && ((4 * hi) % K) == 0) r.x0 = *reinterpret_cast<const float4*>(xa + (pt_ix)iD * C + 4 * piece);
    if (NX >= 2 && ((4 * hi) % K) == 0) r.x1 = *reinterpret_cast<const float4*>(xb + (pt_ix)iD * C + 4 * piece);
    // NP: the p1 values of the point's K pairs are 3 K contiguous floats: the first 3 K / 4 lanes of the point's first lane row fetch them
    r.pv = zero;
    if (NP && ((4 * hi) % K) == 0 && lo < 3 * K / 4) r.pv = *reinterpret_cast<const float4*>(p1 + 3 * (pt_ix)iD * K + 4 * lo);
    return r;
}
template <int K, int NX, bool NP = false>
__device__ __forceinline__ void pt_stage_store(float (*T)[PT_ROWF], const PtStaged<NX>& r, int lo, int hi)
{
    pt_wave_sync();                                                  // the previous tile's reads are done
    *reinterpret_cast<float4*>(&T[4 * hi][4 * lo]) = r.k0;
    *reinterpret_cast<float4*>(&T[4 * hi + 1][4 * lo]) = r.k1;
    *reinterpret_cast<float4*>(&T[4 * hi + 2][4 * lo]) = r.k2;
    *reinterpret_cast<float4*>(&T[4 * hi + 3][4 * lo]) = r.k3;
    if (((4 * hi) % K) == 0) {
        if (NX >= 1) *reinterpret_cast<float4*>(&T[16 + (4 * hi) / K][4 * lo]) = r.x0;
        if (NX >= 2) *reinterpret_cast<float4*>(&T[18 + (4 * hi) / K][4 * lo]) = r.x1;
        if (NP && lo < 3 * K / 4) *reinterpret_cast<float4*>(&T[20][12 * hi + 4 * lo]) = r.pv;      // row 20: [slot][3], slot = 4 hi .. of this point
    }
    pt_wave_sync();
}

// ---- p chain: p_r, p0 and the sums of BN_p (lane = pair) ------------------------------------------------------------------------------
// partial row: p0 [3] | p0^2 [3] | p_r [3] | p0[a] p_r[b] [9]   (the last two feed the Linear(3,3) gradient without another pass, pchain_bwd)
__global__ __launch_bounds__(PT_NARROW_BLOCK) void pt_pchain_kernel(long long npairs, CblFastDiv dvK, const float* __restrict__ xyz, const int* __restrict__ idx,
                                                                    const float* __restrict__ Wp, const float* __restrict__ bp, float* __restrict__ p_r,
                                                                    float* __restrict__ p0, float* __restrict__ partial, const float* __restrict__ cst_eval,
                                                                    float* __restrict__ p1)
{
    // cst_eval (evaluation mode: BN_p's constants are known before the pass): p1 = ReLU(BN_p(p0)) is written here and no statistics pass follows
    __shared__ float red[PT_NARROW_BLOCK / 64][18];
    float w[9], b[3], acc[18];
    float esc[3] = {0.f, 0.f, 0.f}, esh[3] = {0.f, 0.f, 0.f};
    if (cst_eval) {
#pragma unroll
        for (int t = 0; t < 3; t++) { esc[t] = cst_eval[PT_CST_P + t]; esh[t] = cst_eval[PT_CST_P + 4 + t]; }
    }
#pragma unroll
    for (int t = 0; t < 9; t++) w[t] = Wp[t];
#pragma unroll
    for (int t = 0; t < 3; t++) b[t] = bp[t];
#pragma unroll
    for (int t = 0; t < 18; t++) acc[t] = 0.f;
    for (long long p = (long long)blockIdx.x * PT_NARROW_BLOCK + threadIdx.x; p < npairs; p += (long long)gridDim.x * PT_NARROW_BLOCK) {
        const unsigned i = cbl_fastdiv((unsigned)p, dvK);
        const int j = idx[p];
        float r[3], q[3];
#pragma unroll
        for (int t = 0; t < 3; t++) r[t] = xyz[3 * (size_t)j + t] - xyz[3 * (size_t)i + t];
#pragma unroll
        for (int a = 0; a < 3; a++) q[a] = fmaf(w[3 * a + 2], r[2], fmaf(w[3 * a + 1], r[1], fmaf(w[3 * a], r[0], b[a])));
#pragma unroll
        for (int t = 0; t < 3; t++) { p_r[3 * p + t] = r[t]; p0[3 * p + t] = q[t]; acc[t] += q[t]; acc[3 + t] = fmaf(q[t], q[t], acc[3 + t]); acc[6 + t] += r[t]; }
        if (cst_eval) {
#pragma unroll
            for (int t = 0; t < 3; t++) p1[3 * p + t] = fmaxf(fmaf(q[t], esc[t], esh[t]), 0.f);
        }
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int t = 0; t < 3; t++) acc[9 + 3 * a + t] = fmaf(q[a], r[t], acc[9 + 3 * a + t]);
    }
#pragma unroll
    for (int t = 0; t < 18; t++) { const float s = pt_wave_sum(acc[t]); if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][t] = s; }
    __syncthreads();
    if (threadIdx.x < 18) {
        float s = 0.f;
        for (int wv = 0; wv < PT_NARROW_BLOCK / 64; wv++) s += red[wv][threadIdx.x];
        partial[(size_t)blockIdx.x * 18 + threadIdx.x] = s;
    }
}

// ---- BatchNorm finalize: partial rows -> scale / shift / mean / invstd per channel (+ running statistics) ------------------------------
// Workgroup b owns columns 16 b .. 16 b + 15, its 1024 threads = 16 columns x 64 row slices; sums in double, fixed order.
//   column c < n_bn : BatchNorm channel c from the sums at partial[row * stride + off0 + c] and [.. off1 + c]
//   column c < n_raw: the plain sum of partial[row * stride + c], kept at raw_out[c] (forward sums the backward pass needs)
constexpr int PT_FIN_THREADS = 1024;
__global__ __launch_bounds__(PT_FIN_THREADS) void pt_bn_finalize_kernel(int nrows, int stride, const float* __restrict__ partial, int n_bn, int off0, int off1,
                                                                        long long rows, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                        float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                        long long* __restrict__ num_batches_tracked, float* __restrict__ cst, int cst_stride,
                                                                        int n_raw, float* __restrict__ raw_out)
{
    __shared__ double red[64][16][3];
    if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) num_batches_tracked[0] += 1;
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int r = sl; r < nrows; r += 64) {
        const float* row = partial + (size_t)r * stride;
        if (c < n_bn) { a0 += (double)row[off0 + c]; a1 += (double)row[off1 + c]; }
        if (c < n_raw) a2 += (double)row[c];
    }
    red[sl][cl][0] = a0; red[sl][cl][1] = a1; red[sl][cl][2] = a2;
    __syncthreads();
    double tot = 0.0;
    if (sl < 3) for (int j = 0; j < 64; j++) tot += red[j][cl][sl];      // slice t totals quantity t of the column
    __syncthreads();
    if (sl < 3) red[0][cl][sl] = tot;
    __syncthreads();
    if (sl == 0) {
        if (c < n_raw) raw_out[c] = (float)red[0][cl][2];
        if (c < n_bn) {
            const double mu = red[0][cl][0] / (double)rows;
            double var = red[0][cl][1] / (double)rows - mu * mu;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)eps));
            const float scale = gamma[c] * invstd;
            cst[c] = scale;
            cst[cst_stride + c] = beta[c] - (float)mu * scale;
            cst[2 * cst_stride + c] = (float)mu;
            cst[3 * cst_stride + c] = invstd;
            if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
            if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(rows > 1 ? var * (double)rows / (double)(rows - 1) : var);
        }
    }
}

// BatchNorm backward finalize: S1 = sum dy, S2 = sum dy xhat  ->  dx = A1 dy + A2 x + A3;  d gamma = S2, d beta = S1
// (dx = gamma invstd (dy - S1/N - xhat S2/N), xhat = (x - mean) invstd).   lin_bias_grad (optional): the gradient of a bias that feeds this
// BatchNorm directly, sum over rows of dx = A1 S1 + A2 sum(x) + N A3 — analytically 0, returned as the rounding noise the unfused layer returns
__global__ __launch_bounds__(PT_FIN_THREADS) void pt_bn_bwd_finalize_kernel(int nrows, int stride, int off0, int off1, int Cn, long long rows,
                                                                            const float* __restrict__ partial, const float* __restrict__ gamma,
                                                                            const float* __restrict__ cst, int cst_stride, float* __restrict__ bc, int bc_stride,
                                                                            float* __restrict__ g_gamma, float* __restrict__ g_beta,
                                                                            const float* __restrict__ raw_x, float* __restrict__ lin_bias_grad)
{
    __shared__ double red[64][16][2];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
    double a0 = 0.0, a1 = 0.0;
    if (c < Cn)
        for (int r = sl; r < nrows; r += 64) { a0 += (double)partial[(size_t)r * stride + off0 + c]; a1 += (double)partial[(size_t)r * stride + off1 + c]; }
    red[sl][cl][0] = a0; red[sl][cl][1] = a1;
    __syncthreads();
    double tot = 0.0;
    if (sl < 2) for (int j = 0; j < 64; j++) tot += red[j][cl][sl];
    __syncthreads();
    if (sl < 2) red[0][cl][sl] = tot;
    __syncthreads();
    if (sl == 0 && c < Cn) {
        const double s1 = red[0][cl][0], s2 = red[0][cl][1];
        const double mean = (double)cst[2 * cst_stride + c], invstd = (double)cst[3 * cst_stride + c];
        const double A1 = (double)gamma[c] * invstd, m1 = s1 / (double)rows, m2 = s2 / (double)rows;
        const double A2 = -A1 * m2 * invstd, A3 = A1 * (m2 * invstd * mean - m1);
        bc[c] = (float)A1; bc[bc_stride + c] = (float)A2; bc[2 * bc_stride + c] = (float)A3;
        g_gamma[c] = (float)s2; g_beta[c] = (float)s1;
        if (lin_bias_grad) lin_bias_grad[c] = (float)(A1 * s1 + A2 * (double)raw_x[c] + (double)rows * A3);
    }
}

// plain column sums of partial rows (parameter gradients): out[seg.dst + t] = sum over rows of partial[row * stride + off + t]
struct PtSumSeg { const float* src; float* dst; int nrows, stride, off, count; };
struct PtSumSegs { PtSumSeg s[6]; int n; };
__device__ __forceinline__ void pt_sum_rows_body(const PtSumSegs& segs)
{
    __shared__ double red[64][16];
    int b = blockIdx.x;
    for (int q = 0; q < segs.n; q++) {
        const PtSumSeg sg = segs.s[q];
        const int nb = (sg.count + 15) / 16;
        if (b >= nb) { b -= nb; continue; }
        const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4, c = b * 16 + cl;
        double a = 0.0;
        if (c < sg.count) for (int r = sl; r < sg.nrows; r += 64) a += (double)sg.src[(size_t)r * sg.stride + sg.off + c];
        red[sl][cl] = a;
        __syncthreads();
        if (sl == 0 && c < sg.count) { double s = 0.0; for (int j = 0; j < 64; j++) s += red[j][cl]; sg.dst[c] = (float)s; }
        return;
    }
}

__global__ __launch_bounds__(PT_FIN_THREADS) void pt_sum_rows_kernel(PtSumSegs segs) { pt_sum_rows_body(segs); }

// ---- statistics finalize inside the CONSUMER pass ------------------------------------------------------------------------------------
// The narrow BatchNorms (BN_p: 3 channels, BN_g: G <= 8) have partial rows of <= 32 floats: instead of a one-workgroup finalize launch between the
// producer and the consumer (5 - 6 us of kernel + the launch boundary, four times per layer and direction pair), every workgroup of the consumer sums
// the producer's partial rows itself — in double, in one fixed order, so all workgroups hold the same bits — and workgroup 0 also writes what the later
// passes read (consts, running statistics, parameter gradients).  BN_c (2 C = 128 columns x 512 rows per workgroup) keeps its finalize launch.
struct PtFin {
    const float* partial; int nrows, stride; long long rows;
    const float* gamma; const float* beta; float eps, momentum;
    float* running_mean; float* running_var; long long* num_batches; float* consts;
};
// column sums [0, ncols <= PITCH <= 32) of partial (nrows x stride): a thread fetches one ROW (independent 8-byte loads, one memory round trip per NT rows —
// a thread that walks a column pays a dependent round trip per row it owns: measured 8 - 17 us per workgroup with every workgroup of the launch reading the
// same lines), parks it in LDS, and thread (column t % 32, slice t / 32) sums its NT / 32 rows of the chunk from there in double; chunks and slices in a fixed
// order.  Rows and the row stride are 8-byte aligned (every partial-row layout of this file is: strides 18, 16, 8, 88, 28).
// scratch: NT * PITCH floats + NT + 32 doubles of LDS; returns the totals (valid for every thread after the call's barrier)
template <int NT, int PITCH>
__device__ __forceinline__ const double* pt_colsum_block(const float* __restrict__ partial, int nrows, int stride, int ncols, float* scratch_f)
{
    constexpr int S = NT / 32, NQ = PITCH / 2;
    float* rows = scratch_f;                                          // [NT][PITCH]
    double* scratch = reinterpret_cast<double*>(scratch_f + NT * PITCH);
    const int c = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int nq = (ncols + 1) >> 1;
    double acc = 0.0;
    for (int base = 0; base < nrows; base += NT) {
        const int r = base + threadIdx.x;
        float2 v[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) v[q] = (q < nq && r < nrows) ? *reinterpret_cast<const float2*>(partial + (size_t)r * stride + 2 * q) : make_float2(0.f, 0.f);
        if (base) __syncthreads();                                    // the previous chunk is consumed
#pragma unroll
        for (int q = 0; q < NQ; q++) *reinterpret_cast<float2*>(rows + threadIdx.x * PITCH + 2 * q) = v[q];
        __syncthreads();
        if (c < ncols) {
            const int lim = min(NT, nrows - base);
            for (int rr = sl; rr < lim; rr += S) acc += (double)rows[rr * PITCH + c];
        }
    }
    scratch[threadIdx.x] = acc;
    __syncthreads();
    if ((int)threadIdx.x < ncols) { double t = 0.0; for (int j = 0; j < S; j++) t += scratch[j * 32 + threadIdx.x]; scratch[NT + threadIdx.x] = t; }
    __syncthreads();
    return scratch + NT;
}
// forward: scale / shift of n_bn <= 16 channels (sums at columns [0, n_bn), squares at [off1, off1 + n_bn)) -> fout[c], fout[16 + c] (LDS floats behind the doubles);
// workgroup 0: consts (scale, shift, mean, invstd at cst_off + {0, 1, 2, 3} cst_stride), the raw sums of the first n_raw columns at raw_off, running statistics.
// The caller reads fout and passes a barrier before the scratch memory is used for anything else.
template <int NT, int PITCH>
__device__ __forceinline__ const float* pt_fin_forward(const PtFin& f, int n_bn, int off1, int ncols, int cst_off, int cst_stride, int n_raw, int raw_off, float* scratch)
{
    const double* tot = pt_colsum_block<NT, PITCH>(f.partial, f.nrows, f.stride, ncols, scratch);
    float* fout = scratch;                                            // the row chunk is consumed: [2][16] floats over it
    if ((int)threadIdx.x < n_bn) {
        const int c = threadIdx.x;
        const double mu = tot[c] / (double)f.rows;
        double var = tot[off1 + c] / (double)f.rows - mu * mu;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
        const float scale = f.gamma[c] * invstd, shift = f.beta[c] - (float)mu * scale;
        fout[c] = scale; fout[16 + c] = shift;
        if (blockIdx.x == 0) {
            float* cst = f.consts + cst_off;
            cst[c] = scale; cst[cst_stride + c] = shift; cst[2 * cst_stride + c] = (float)mu; cst[3 * cst_stride + c] = invstd;
            if (f.running_mean) f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * (float)mu;
            if (f.running_var) f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * (float)(f.rows > 1 ? var * (double)f.rows / (double)(f.rows - 1) : var);
        }
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < n_raw) f.consts[raw_off + threadIdx.x] = (float)tot[threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 0 && f.num_batches) f.num_batches[0] += 1;
    __syncthreads();
    return fout;
}

// backward: d x = A1 d y + A2 x + A3 of Cn <= 8 channels from the sums S1 = sum d y (columns [0, Cn)) and S2 = sum d y xhat ([Cn, 2 Cn)) -> fout[c], fout[8 + c],
// fout[16 + c]; workgroup 0: d gamma = S2, d beta = S1, and (lin_bias_grad) the gradient of a bias that feeds the BatchNorm directly (pt_bn_bwd_finalize_kernel)
struct PtFinBwd {
    const float* partial; int nrows, stride; long long rows;
    const float* gamma; const float* cst; int cst_stride; const float* raw_x; float* g_gamma; float* g_beta; float* lin_bias_grad;
};
template <int NT, int PITCH>
__device__ __forceinline__ const float* pt_fin_backward(const PtFinBwd& f, int Cn, float* scratch)
{
    const double* tot = pt_colsum_block<NT, PITCH>(f.partial, f.nrows, f.stride, 2 * Cn, scratch);
    float* fout = scratch;                                            // the row chunk is consumed: [3][8] floats over it
    if ((int)threadIdx.x < Cn) {
        const int c = threadIdx.x;
        const double s1 = tot[c], s2 = tot[Cn + c];
        const double mean = (double)f.cst[2 * f.cst_stride + c], invstd = (double)f.cst[3 * f.cst_stride + c];
        const double A1 = (double)f.gamma[c] * invstd, m1 = s1 / (double)f.rows, m2 = s2 / (double)f.rows;
        const double A2 = -A1 * m2 * invstd, A3 = A1 * (m2 * invstd * mean - m1);
        fout[c] = (float)A1; fout[8 + c] = (float)A2; fout[16 + c] = (float)A3;
        if (blockIdx.x == 0) {
            f.g_gamma[c] = (float)s2; f.g_beta[c] = (float)s1;
            if (f.lin_bias_grad) f.lin_bias_grad[c] = (float)(A1 * s1 + A2 * (double)f.raw_x[c] + (double)f.rows * A3);
        }
    }
    __syncthreads();
    return fout;
}

// ---- the C-wide pair chain ------------------------------------------------------------------------------------------------------------
// Per-lane constant of the pe product: element (channel 16 ct + lo, d = hi) of [W3C | b3C]; the A operand of the pair-major form
// (D[channel][pair]) and the B operand of the channel-major form (D[pair][channel]) are the same lane mapping.
template <int C> struct PtPe { float w[C / 16]; };
template <int C> __device__ __forceinline__ PtPe<C> pt_pe_load(const float* __restrict__ W3C, const float* __restrict__ b3C, int lo, int hi)
{
    PtPe<C> r;
#pragma unroll
    for (int ct = 0; ct < C / 16; ct++) r.w[ct] = hi < 3 ? W3C[3 * (16 * ct + lo) + hi] : b3C[16 * ct + lo];
    return r;
}

// ---- BN_c statistics of w (pair-major); writes p1 ---------------------------------------------------------------------------------------
// partial row: sum w [C] | sum w^2 [C]
template <int C, int K>
__global__ __launch_bounds__(PT_BLOCK) void pt_wstats_kernel(int n, const int* __restrict__ order, const float* __restrict__ xq, const float* __restrict__ xk,
                                                             const int* __restrict__ idx, const float* __restrict__ p0, PtFin fin_p,
                                                             const float* __restrict__ W3C, const float* __restrict__ b3C, float* __restrict__ p1,
                                                             float* __restrict__ partial)
{
    constexpr int CT = C / 16;
    __shared__ __attribute__((aligned(16))) float tile[PT_WPB][PT_TROWS][PT_ROWF];
    const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4, wave = threadIdx.x >> 6;
    float (*T)[PT_ROWF] = tile[wave];
    const PtPe<C> pe = pt_pe_load<C>(W3C, b3C, lo, hi);
    float psc = 0.f, psh = 1.f;                                      // BN_p's scale / shift of channel hi (hi = 3: the 1 that multiplies the bias)
    float s0[CT][4], s1[CT][4];
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
#pragma unroll
        for (int v = 0; v < 4; v++) { s0[ct][v] = 0.f; s1[ct][v] = 0.f; }
    const unsigned ntiles = (unsigned)(((long long)n * K + 15) / 16);
    struct S1 { int4 j; float p0v; };
    pt_pipeline(ntiles,
        [&](unsigned tl) { return pt_stage0<K>(tl, ntiles, n, order, lo, hi); },
        [&](const PtS0& a) { S1 b; b.j = *reinterpret_cast<const int4*>(idx + a.pD); b.p0v = hi < 3 ? p0[3 * (pt_ix)a.pA + hi] : 0.f; return b; },
        [&](const PtS0& a, const S1& b) { return pt_stage_rows<C, K, 1>(xk, b.j, xq, nullptr, a.iD, lo, hi); },
        [&](const PtS0& a, const S1& b, const PtStaged<1>& r) {
            pt_stage_store<K, 1>(T, r, lo, hi);
            // p1 of (pair, d = hi): this lane's B operand; hi = 3 carries the 1 that multiplies the bias
            const float p1x = fmaxf(fmaf(b.p0v, psc, psh), 0.f);
            if (hi < 3 && a.vA) p1[3 * (pt_ix)a.pA + hi] = p1x;
#pragma unroll
            for (int ct = 0; ct < CT; ct++) {
                const float4 kv = *reinterpret_cast<const float4*>(&T[lo][16 * ct + 4 * hi]), qv = *reinterpret_cast<const float4*>(&T[16 + lo / K][16 * ct + 4 * hi]);
                pt_f32x4 w = pt_vec4(kv.x - qv.x, kv.y - qv.y, kv.z - qv.z, kv.w - qv.w);
                w = pt_mfma(pe.w[ct], p1x, w);
                if (a.vA) {
#pragma unroll
                    for (int v = 0; v < 4; v++) { s0[ct][v] += w[v]; s1[ct][v] = fmaf(w[v], w[v], s1[ct][v]); }
                }
            }
        },
        [&]() {
            // BN_p's constants from the p chain's partial rows (every workgroup the same bits; workgroup 0 keeps them in consts and moves the running statistics)
            const float* fo = pt_fin_forward<PT_BLOCK, 18>(fin_p, 3, 3, 18, PT_CST_P, 4, 18, PT_FS_P, &tile[0][0][0]);
            if (hi < 3) { psc = fo[hi]; psh = fo[16 + hi]; }
            __syncthreads();                                         // the scratch becomes the waves' tiles
        });
    __syncthreads();                                                 // the tiles are done with: their LDS carries the workgroup's partial row now
    float* red = &tile[0][0][0];
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const float a = pt_row_sum(s0[ct][v]), b = pt_row_sum(s1[ct][v]);
            if (lo == 0) { red[wave * 2 * C + 16 * ct + 4 * hi + v] = a; red[wave * 2 * C + C + 16 * ct + 4 * hi + v] = b; }
        }
    __syncthreads();
    if (threadIdx.x < 2 * C) {
        float s = 0.f;
        for (int wv = 0; wv < PT_WPB; wv++) s += red[wv * 2 * C + threadIdx.x];
        partial[(size_t)blockIdx.x * (2 * C) + threadIdx.x] = s;
    }
}

// ---- w2 = Wa . ReLU(BN_c(w)) + ba (pair-major, the C -> G contraction on the matrix cores) + the sums of BN_g ----------------------------
// partial row: sum w2 [G] | sum w2^2 [G]
template <int C, int K>
__global__ __launch_bounds__(PT_BLOCK) void pt_w2_kernel(int n, const int* __restrict__ order, const float* __restrict__ xq, const float* __restrict__ xk,
                                                         const int* __restrict__ idx, const float* __restrict__ p1, const float* __restrict__ cst,
                                                         const float* __restrict__ W3C, const float* __restrict__ b3C, const float* __restrict__ Wa,
                                                         const float* __restrict__ ba, float* __restrict__ w2, float* __restrict__ partial)
{
    constexpr int CT = C / 16, G = C / 8;
    __shared__ float tile[PT_WPB][PT_TROWS][PT_ROWF];
    __shared__ float red[PT_WPB][2 * G];
    const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4, wave = threadIdx.x >> 6;
    float (*T)[PT_ROWF] = tile[wave];
    const PtPe<C> pe = pt_pe_load<C>(W3C, b3C, lo, hi);
    float4 sc[CT], sh[CT], wb[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        sc[ct] = *reinterpret_cast<const float4*>(cst + PT_CST_C + 16 * ct + 4 * hi);
        sh[ct] = *reinterpret_cast<const float4*>(cst + PT_CST_C + 64 + 16 * ct + 4 * hi);
        wb[ct] = lo < G ? *reinterpret_cast<const float4*>(Wa + (size_t)lo * C + 16 * ct + 4 * hi) : make_float4(0.f, 0.f, 0.f, 0.f);   // B[k = hi][col = g]
    }
    const float bias = lo < G ? ba[lo] : 0.f;
    float t0 = 0.f, t1 = 0.f;
    const unsigned ntiles = (unsigned)(((long long)n * K + 15) / 16);
    struct S1 { int4 j; float p1x; };
    pt_pipeline(ntiles,
        [&](unsigned tl) { return pt_stage0<K>(tl, ntiles, n, order, lo, hi); },
        [&](const PtS0& a) { S1 b; b.j = *reinterpret_cast<const int4*>(idx + a.pD); b.p1x = hi < 3 ? p1[3 * (pt_ix)a.pA + hi] : 1.f; return b; },
        [&](const PtS0& a, const S1& b) { return pt_stage_rows<C, K, 1>(xk, b.j, xq, nullptr, a.iD, lo, hi); },
        [&](const PtS0& a, const S1& b, const PtStaged<1>& r) {
            pt_stage_store<K, 1>(T, r, lo, hi);
            pt_f32x4 o0 = pt_vec4(bias, bias, bias, bias), o1 = pt_vec4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int ct = 0; ct < CT; ct++) {
                const float4 kv = *reinterpret_cast<const float4*>(&T[lo][16 * ct + 4 * hi]), qv = *reinterpret_cast<const float4*>(&T[16 + lo / K][16 * ct + 4 * hi]);
                pt_f32x4 w = pt_vec4(kv.x - qv.x, kv.y - qv.y, kv.z - qv.z, kv.w - qv.w);
                w = pt_mfma(pe.w[ct], b.p1x, w);
                o0 = pt_mfma(fmaxf(fmaf(w[0], sc[ct].x, sh[ct].x), 0.f), wb[ct].x, o0);
                o1 = pt_mfma(fmaxf(fmaf(w[1], sc[ct].y, sh[ct].y), 0.f), wb[ct].y, o1);
                o0 = pt_mfma(fmaxf(fmaf(w[2], sc[ct].z, sh[ct].z), 0.f), wb[ct].z, o0);
                o1 = pt_mfma(fmaxf(fmaf(w[3], sc[ct].w, sh[ct].w), 0.f), wb[ct].w, o1);
            }
            // D[pair slot 4 hi + v][g = lo]
            if (lo < G && a.vD) {
#pragma unroll
                for (int v = 0; v < 4; v++) { const float x = o0[v] + o1[v]; w2[(pt_ix)(a.pD + v) * G + lo] = x; t0 += x; t1 = fmaf(x, x, t1); }
            }
        });
    t0 += pt_xor16(t0); t0 += pt_xor32(t0); t1 += pt_xor16(t1); t1 += pt_xor32(t1);
    if (hi == 0 && lo < G) { red[wave][lo] = t0; red[wave][G + lo] = t1; }
    __syncthreads();
    if (threadIdx.x < 2 * G) {
        float s = 0.f;
        for (int wv = 0; wv < PT_WPB; wv++) s += red[wv][threadIdx.x];
        partial[(size_t)blockIdx.x * (2 * G) + threadIdx.x] = s;
    }
}

// ---- softmax + aggregation (channel-major): a = softmax over K of Linear(G,G)(ReLU(BN_g(w2))); out[i, c] = sum_k (x_v[j] + pe) a[.., c % G] ----------
// !BWD (forward): BN_g's constants come from the w2 pass's partial rows in the prologue (fin_g.nrows > 0; evaluation mode: from cst), the logits of
// (slot lo, g = hi | hi + 4) are formed pair-major — the K slots of a point are K adjacent lanes of a row, so the softmax is two DPP folds —, pass through a
// [slot][g] table in the wave's LDS tile into the channel-major operand a[slot 4 hi + v][lo % G], and are stored for the backward pass from that table
// (16 bytes per lane).  The separate softmax launch of round 4 read w2 and wrote a (2 x 21 MB at (40960, 16, 64)) for 12 us + a launch boundary.
// BWD: d logits = a (d a - sum_k a d a) with d a[k, g] = sum over c = g (mod G) of d out[c] (x_v[j, c] + pe[c]) — and, in the same pass (round 6; a launch of
// its own until then: 63 MB of narrow traffic for 19 us), the narrow backward behind it: pre = (y > 0) Wb^T d logits with y = BN_g(w2) (d w2 BEFORE BN_g's
// backward, written in place of d logits, which no other pass reads), its two BatchNorm sums, d Wb, d bb.  The G logit gradients of a slot sit in the G lanes
// lo < G of a lane row; they pass through the wave's [slot][g] table, and lane g' of the row forms column g' of the product and owns the sums of channel g'.
//   partial row (BWD): S1 [G] | S2 [G] | d Wb [G][G] | d bb [G]
template <int C, int K, bool BWD>
__global__ __launch_bounds__(PT_BLOCK) void pt_agg_kernel(int n, const int* __restrict__ order, const float* __restrict__ xv, const int* __restrict__ idx,
                                                          const float* __restrict__ p1, const float* __restrict__ W3C, const float* __restrict__ b3C,
                                                          float* __restrict__ a, float* __restrict__ out, const float* __restrict__ gout,
                                                          float* __restrict__ glogit, const float* __restrict__ w2, const float* __restrict__ Wb,
                                                          const float* __restrict__ bb, const float* __restrict__ cst, PtFin fin_g, float* __restrict__ partial)
{
    constexpr int CT = C / 16, G = C / 8, NX = BWD ? 1 : 0, WN = 3 * G + G * G;
    constexpr int TILEF = PT_TROWS * PT_ROWF;
    static_assert(PT_WPB * WN <= PT_WPB * TILEF, "the backward's partial rows lie over the tiles");
    __shared__ __attribute__((aligned(16))) float lds[PT_WPB * TILEF + 96];     // the waves' tiles | scale [8], shift [8], Wb [G][G], bb [8] (forward) / invstd, -mean invstd (backward) — ONE array (pt_w2_bwd_kernel)
    const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4, wave = threadIdx.x >> 6;
    float (*T)[PT_ROWF] = reinterpret_cast<float (*)[PT_ROWF]>(lds + wave * TILEF);
    float* AT = &T[18][0];                                           // the tile's narrow [slot][g] table (forward: softmax weights, backward: d logits): rows 18 - 19, which no staging of this pass uses
    float* ctab = lds + PT_WPB * TILEF;
    const PtPe<C> pe = pt_pe_load<C>(W3C, b3C, lo, hi);
    const int o0 = hi < G ? hi : 0, o1 = G == 8 ? hi + 4 : 0;
    // BWD: this lane's channel g' = lo % G of the narrow backward: BN_g constants, column g' of Wb, and the sums it owns
    float nsc = 0.f, nsh = 0.f, nis = 0.f, nnm = 0.f, wcol[G], nacc[G], ns1 = 0.f, ns2 = 0.f, nsb = 0.f;
#pragma unroll
    for (int g = 0; g < G; g++) { wcol[g] = 0.f; nacc[g] = 0.f; }
    if (BWD) {
        const int gq = lo % G;
        nsc = cst[PT_CST_G + gq]; nsh = cst[PT_CST_G + 8 + gq]; nis = cst[PT_CST_G + 24 + gq]; nnm = -cst[PT_CST_G + 16 + gq] * nis;
#pragma unroll
        for (int g = 0; g < G; g++) wcol[g] = Wb[g * G + gq];
    }
    const unsigned ntiles = (unsigned)(((long long)n * K + 15) / 16);
    struct S1 { int4 j; float p1x; float av[4]; float4 wlo, whi; };
    pt_pipeline(ntiles,
        [&](unsigned tl) { return pt_stage0<K>(tl, ntiles, n, order, lo, hi); },
        [&](const PtS0& t) {
            S1 b;
            b.j = *reinterpret_cast<const int4*>(idx + t.pD);
            b.p1x = hi < 3 ? p1[3 * (pt_ix)t.pA + hi] : 1.f;                     // A[pair slot lo][d = hi]
            b.wlo = b.whi = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int v = 0; v < 4; v++) b.av[v] = 0.f;
            if (BWD) {
#pragma unroll
                for (int v = 0; v < 4; v++) b.av[v] = a[(pt_ix)(t.pD + v) * G + (lo % G)];
                b.wlo = make_float4(w2[(pt_ix)t.pD * G + (lo % G)], w2[(pt_ix)(t.pD + 1) * G + (lo % G)], w2[(pt_ix)(t.pD + 2) * G + (lo % G)], w2[(pt_ix)(t.pD + 3) * G + (lo % G)]);
            } else {                                                             // w2 of pair slot lo, all G values
                b.wlo = *reinterpret_cast<const float4*>(w2 + (pt_ix)t.pA * G);
                if (G == 8) b.whi = *reinterpret_cast<const float4*>(w2 + (pt_ix)t.pA * G + 4);
            }
            return b;
        },
        [&](const PtS0& t, const S1& b) { return pt_stage_rows<C, K, NX>(xv, b.j, gout, nullptr, t.iD, lo, hi); },
        [&](const PtS0& t, const S1& b, const PtStaged<NX>& r) {
            pt_stage_store<K, NX>(T, r, lo, hi);
            float av[4] = {b.av[0], b.av[1], b.av[2], b.av[3]};
            if (!BWD) {
                const float xin[8] = {b.wlo.x, b.wlo.y, b.wlo.z, b.wlo.w, b.whi.x, b.whi.y, b.whi.z, b.whi.w};
                int z = 0;
                asm volatile("" : "+v"(z));                           // opaque per tile: the table reads stay LDS reads inside the loop
                const float* ct = ctab + z;
                float l0 = ct[80 + o0], l1 = ct[80 + o1];
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const float x = fmaxf(fmaf(xin[g], ct[g], ct[8 + g]), 0.f);
                    l0 = fmaf(ct[16 + o0 * G + g], x, l0);
                    if (G == 8) l1 = fmaf(ct[16 + o1 * G + g], x, l1);
                }
                // (v_exp_f32 / v_rcp_f32 forms measured: no faster — the pass is not bound by these)
                const float e0 = expf(l0 - pt_group_max<K>(l0));
                const float a0 = e0 / pt_group_sum<K>(e0);
                if (hi < G) AT[lo * G + hi] = a0;
                if (G == 8) {
                    const float e1 = expf(l1 - pt_group_max<K>(l1));
                    AT[lo * G + hi + 4] = e1 / pt_group_sum<K>(e1);
                }
                pt_wave_sync();
#pragma unroll
                for (int v = 0; v < 4; v++) av[v] = AT[(4 * hi + v) * G + (lo % G)];
                // the four slots of this lane row are 4 G consecutive floats of `a`: lane lo < G stores 16 bytes of them
                if (lo < G && t.vD) *reinterpret_cast<float4*>(a + (pt_ix)t.pD * G + 4 * lo) = *reinterpret_cast<const float4*>(&AT[4 * hi * G + 4 * lo]);
            }
            pt_f32x4 val[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ct++) {
                val[ct] = pt_vec4(T[4 * hi][16 * ct + lo], T[4 * hi + 1][16 * ct + lo], T[4 * hi + 2][16 * ct + lo], T[4 * hi + 3][16 * ct + lo]);
                val[ct] = pt_mfma(b.p1x, pe.w[ct], val[ct]);                       // x_v[j] + pe of (slot 4 hi + v, channel 16 ct + lo)
            }
            if (!BWD) {
#pragma unroll
                for (int ct = 0; ct < CT; ct++) {
                    float o = fmaf(av[3], val[ct][3], fmaf(av[2], val[ct][2], fmaf(av[1], val[ct][1], av[0] * val[ct][0])));
                    o = pt_point_sum<K>(o);
                    if (t.vD && (K == 16 ? hi == 0 : (hi & 1) == 0)) out[(pt_ix)t.iD * C + 16 * ct + lo] = o;
                }
            } else {
                float ga[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ct = 0; ct < CT; ct++) {
                    const float go = T[16 + (4 * hi) / K][16 * ct + lo];
#pragma unroll
                    for (int v = 0; v < 4; v++) ga[v] = fmaf(go, val[ct][v], ga[v]);
                }
                float dot = 0.f;
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    ga[v] += pt_row_ror8(ga[v]);                                     // the lanes lo = g (mod G)
                    if (G == 4) ga[v] += pt_row_ror4(ga[v]);
                    dot = fmaf(av[v], ga[v], dot);
                }
                dot = pt_point_sum<K>(dot);
                // d logits of (slot 4 hi + v, g = lo) in the lanes lo < G -> the wave's [slot][g] table; 0 for slots past the end (they add nothing below)
                float gl[4];
#pragma unroll
                for (int v = 0; v < 4; v++) { gl[v] = t.vD ? av[v] * (ga[v] - dot) : 0.f; if (lo < G) AT[(4 * hi + v) * G + lo] = gl[v]; }
                pt_wave_sync();
                const float w2v[4] = {b.wlo.x, b.wlo.y, b.wlo.z, b.wlo.w};
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    float row[G];
#pragma unroll
                    for (int q = 0; q < G / 4; q++) {
                        const float4 x = *reinterpret_cast<const float4*>(&AT[(4 * hi + v) * G + 4 * q]);
                        row[4 * q] = x.x; row[4 * q + 1] = x.y; row[4 * q + 2] = x.z; row[4 * q + 3] = x.w;
                    }
                    float sacc = 0.f;
#pragma unroll
                    for (int g = 0; g < G; g++) sacc = fmaf(wcol[g], row[g], sacc);
                    const float y = fmaf(w2v[v], nsc, nsh);
                    const float d = y > 0.f ? sacc : 0.f;
                    if (lo < G && t.vD) glogit[(pt_ix)(t.pD + v) * G + lo] = d;            // `pre`: what the reduce pass takes through BN_g's backward
                    const float ry = fmaxf(y, 0.f), glo = gl[v];         // this lane's own d logit (g = lo % G: every lane of the row holds its channel's value)
                    ns1 += d; ns2 = fmaf(d, fmaf(w2v[v], nis, nnm), ns2); nsb += glo;
#pragma unroll
                    for (int g = 0; g < G; g++) nacc[g] = fmaf(row[g], ry, nacc[g]);          // d Wb[g][g'] += d logit[g] relu(y)[g']
                }
            }
        },
        [&]() {
            if (!BWD) {
                // the narrow constants live in an LDS table and are read where they are used (34 values per lane otherwise: 143 registers, one workgroup per CU)
                if (fin_g.nrows > 0) {
                    const float* fo = pt_fin_forward<PT_BLOCK, 2 * G>(fin_g, G, G, 2 * G, PT_CST_G, 8, G, PT_FS_G, lds);
                    if ((int)threadIdx.x < G) { ctab[threadIdx.x] = fo[threadIdx.x]; ctab[8 + threadIdx.x] = fo[16 + threadIdx.x]; }
                } else if ((int)threadIdx.x < G) { ctab[threadIdx.x] = cst[PT_CST_G + threadIdx.x]; ctab[8 + threadIdx.x] = cst[PT_CST_G + 8 + threadIdx.x]; }
                if ((int)threadIdx.x < G * G) ctab[16 + threadIdx.x] = Wb[threadIdx.x];
                if ((int)threadIdx.x < G) ctab[16 + 64 + threadIdx.x] = bb[threadIdx.x];
                __syncthreads();                                             // table complete; the scratch becomes the waves' tiles
            }
        });
    if (BWD) {
        // the narrow backward's partial row: lanes lo < G of the four lane rows hold channel g' = lo's sums over their slots (lanes lo >= G repeat them: not read)
        __syncthreads();                                             // every wave is done with its tile
        float (*red)[WN] = reinterpret_cast<float (*)[WN]>(lds);
        ns1 = pt_point_sum<16>(ns1); ns2 = pt_point_sum<16>(ns2); nsb = pt_point_sum<16>(nsb);
#pragma unroll
        for (int g = 0; g < G; g++) nacc[g] = pt_point_sum<16>(nacc[g]);
        if (hi == 0 && lo < G) {
            red[wave][lo] = ns1; red[wave][G + lo] = ns2; red[wave][2 * G + G * G + lo] = nsb;
#pragma unroll
            for (int g = 0; g < G; g++) red[wave][2 * G + g * G + lo] = nacc[g];
        }
        __syncthreads();
        for (int tt = threadIdx.x; tt < WN; tt += PT_BLOCK) {
            float sum = 0.f;
            for (int wv = 0; wv < PT_WPB; wv++) sum += red[wv][tt];
            partial[(size_t)blockIdx.x * WN + tt] = sum;
        }
    }
}

// ---- the two C-wide backward passes (channel-major) -----------------------------------------------------------------------------------
// REDUCE: d w2 = BN_g backward of `pre` (written for the later passes); d y1 = mask . Wa^T d w2; sums S1 = sum d y1, S2 = sum d y1 what per channel;
//         d Wa[g, c] = sum over pairs of d w2[g] w1[c] in MFMA accumulators.        partial row: S1 [C] | S2 [C] | d Wa [G][C]
// APPLY:  d w = A1 d y1 + A2 w + A3; d x_q = - sum_k d w; d pe = d w + d out . a; d p1 = W3C^T d pe; d [W3C | b3C] = sum over pairs of d pe (x) [p1, 1].
//         partial row: d W3C [C][3] | d b3C [C]
template <int C, int K, bool APPLY>
__global__ __launch_bounds__(PT_BLOCK) __attribute__((amdgpu_waves_per_eu(PT_BWD_WAVES, PT_BWD_WAVES))) void pt_w2_bwd_kernel(int n, const int* __restrict__ order, const float* __restrict__ xq, const float* __restrict__ xk,
                                                             const int* __restrict__ idx, const float* __restrict__ p1, const float* __restrict__ cst,
                                                             const float* __restrict__ bc, const float* __restrict__ W3C, const float* __restrict__ b3C,
                                                             const float* __restrict__ Wa, const float* __restrict__ w2, const float* __restrict__ pre,
                                                             float* __restrict__ gw2, const float* __restrict__ a, const float* __restrict__ gout,
                                                             float* __restrict__ gxq, float* __restrict__ gp1, float* __restrict__ partial, PtFinBwd fin_g)
{
    constexpr int CT = C / 16, G = C / 8;
    constexpr int W = APPLY ? 4 * C : 2 * C + G * C;
    constexpr int TILEF = PT_TROWS * PT_ROWF;
    constexpr bool TAB = APPLY || PT_BWD_WAVES >= 3;                  // per-channel constants in an LDS table (REDUCE: only where the register budget asks for it)
    constexpr int CTF = TAB ? (5 + G + 3) * C : 0;                    // scale, shift, k1, k2, k3 | Wa [G][C] | (APPLY) W3C transposed [3][C]
    __shared__ __attribute__((aligned(16))) float lds[PT_WPB * (W > TILEF ? W : TILEF) + CTF];    // the waves' staged tiles (then the workgroup's partial row) | CTF — ONE array: a second
                                                                     // __shared__ object makes the compiler drain the prefetched loads before every LDS read
    const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4, wave = threadIdx.x >> 6;
    float (*T)[PT_ROWF] = reinterpret_cast<float (*)[PT_ROWF]>(lds + wave * TILEF);
    const PtPe<C> pe = pt_pe_load<C>(W3C, b3C, lo, hi);
    // APPLY: the per-channel constants (scale, shift, k1..k3, Wa, W3C: 40 values per lane at C = 64) live in an LDS table and are read where they are used
    float* ctab = lds + PT_WPB * (W > TILEF ? W : TILEF);
    float sc[CT], sh[CT], k1[CT], k2[CT], wa0[CT], wa1[CT];           // REDUCE: 24 registers (its table reads measured slower: 57.6 against 54.4 us)
    if (APPLY) {
        for (int c = threadIdx.x; c < C; c += PT_BLOCK) {
            ctab[c] = cst[PT_CST_C + c]; ctab[C + c] = cst[PT_CST_C + 64 + c];
            ctab[2 * C + c] = bc[PT_BC_C + c]; ctab[3 * C + c] = bc[PT_BC_C + 64 + c]; ctab[4 * C + c] = bc[PT_BC_C + 128 + c];
        }
        for (int e = threadIdx.x; e < G * C; e += PT_BLOCK) ctab[5 * C + e] = Wa[e];
        for (int e = threadIdx.x; e < 3 * C; e += PT_BLOCK) ctab[(5 + G) * C + e] = W3C[3 * (e % C) + e / C];     // [d][c]
        __syncthreads();
    } else if (TAB) {
        for (int c = threadIdx.x; c < C; c += PT_BLOCK) {
            const float is = cst[PT_CST_C + 192 + c];
            ctab[c] = cst[PT_CST_C + c]; ctab[C + c] = cst[PT_CST_C + 64 + c]; ctab[2 * C + c] = is; ctab[3 * C + c] = -cst[PT_CST_C + 128 + c] * is;
        }
        for (int e = threadIdx.x; e < G * C; e += PT_BLOCK) ctab[5 * C + e] = Wa[e];
        __syncthreads();
    }
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        const int c = 16 * ct + lo;
        sc[ct] = sh[ct] = k1[ct] = k2[ct] = wa0[ct] = wa1[ct] = 0.f;
        if (!TAB) {
            sc[ct] = cst[PT_CST_C + c]; sh[ct] = cst[PT_CST_C + 64 + c];
            k1[ct] = cst[PT_CST_C + 192 + c];                                    // invstd
            k2[ct] = -cst[PT_CST_C + 128 + c] * k1[ct];                          // - mean invstd
            wa0[ct] = hi < G ? Wa[(size_t)hi * C + c] : 0.f;                     // B[k = g][col = channel]
            wa1[ct] = G == 8 ? Wa[(size_t)(hi + 4) * C + c] : 0.f;
        }
    }
    // BN_g backward constants of the narrow values this lane forms: g = hi, hi + 4 (pair-major operand of d y) and g = lo (operand of d Wa)
    float ga1[3] = {0.f, 0.f, 0.f}, ga2[3] = {0.f, 0.f, 0.f}, ga3[3] = {0.f, 0.f, 0.f};
    // accw: REDUCE d Wa[g][channel] in matrix accumulators (rows g: 8 of 16 used); APPLY d [W3C | b3C][channel][d] as four plain sums per channel block —
    // the f32 matrix instruction runs at the vector rate (32 cycles = 16 v_fma issue slots per 16 x 16 x 4 tile), so a product that uses 4 of its 16 rows
    // costs four times the 16 v_fma that do the same work (round 5: per-tile issue cycles 2 VALU + 32 MFMA explain every pass of this file)
    float s1[CT], s2[CT];
    pt_f32x4 accw[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ct++) { s1[ct] = 0.f; s2[ct] = 0.f; accw[ct] = pt_vec4(0.f, 0.f, 0.f, 0.f); }
    const unsigned ntiles = (unsigned)(((long long)n * K + 15) / 16);
    // level 1: neighbour ids and the narrow values.  REDUCE: `pre` / w2 at (slot lo, g = hi | hi + 4) -> x0..x3 and at (slot 4 hi + v, g = lo) -> u / y;
    // APPLY: d w2 at (slot lo, g = hi | hi + 4) -> x0, x1; [p1, 1] at (slot 4 hi + v, d = lo) -> u; a at (slot 4 hi + v, lo % G) -> y
    struct S1 { int4 j; float p1x, x0, x1, x2, x3; float u[4], y[4]; };
    constexpr int NX = APPLY ? 2 : 1;
    pt_pipeline(ntiles,
        [&](unsigned tl) { return pt_stage0<K>(tl, ntiles, n, order, lo, hi); },
        [&](const PtS0& t) {
            S1 b;
            b.j = *reinterpret_cast<const int4*>(idx + t.pD);
            b.p1x = hi < 3 ? p1[3 * (pt_ix)t.pA + hi] : 1.f;
            b.x0 = b.x1 = b.x2 = b.x3 = 0.f;
            const pt_ix ra = (pt_ix)t.pA * G;
            if (APPLY) {
                if (hi < G) b.x0 = gw2[ra + hi];
                if (G == 8) b.x1 = gw2[ra + hi + 4];
            } else {
                if (hi < G) { b.x0 = pre[ra + hi]; b.x2 = w2[ra + hi]; }
                if (G == 8) { b.x1 = pre[ra + hi + 4]; b.x3 = w2[ra + hi + 4]; }
            }
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const pt_ix rd = (pt_ix)(t.pD + v);
                if (APPLY) { b.u[v] = 0.f; b.y[v] = a[rd * G + (lo % G)]; }
                else { b.u[v] = lo < G ? pre[rd * G + lo] : 0.f; b.y[v] = lo < G ? w2[rd * G + lo] : 0.f; }
            }
            return b;
        },
        [&](const PtS0& t, const S1& b) { return pt_stage_rows<C, K, NX, APPLY>(xk, b.j, xq, gout, t.iD, lo, hi, p1); },
        [&](const PtS0& t, const S1& b, const PtStaged<NX>& r) {
            pt_stage_store<K, NX, APPLY>(T, r, lo, hi);
            // d w2 of (pair slot lo, g = hi / hi + 4): the A operand of d y = d w2 . Wa
            float da0 = 0.f, da1 = 0.f;
            if (APPLY) {
                if (t.vA) { da0 = b.x0; da1 = b.x1; }
            } else {
                const pt_ix ra = (pt_ix)t.pA * G;
                if (t.vA && hi < G) { da0 = fmaf(ga1[0], b.x0, fmaf(ga2[0], b.x2, ga3[0])); gw2[ra + hi] = da0; }
                if (t.vA && G == 8) { da1 = fmaf(ga1[1], b.x1, fmaf(ga2[1], b.x3, ga3[1])); gw2[ra + hi + 4] = da1; }
            }
            // the narrow operands per (slot 4 hi + v): REDUCE d w2[.., g = lo] (A of d Wa), APPLY [p1, 1][.., d = lo] (A of d W3C) and a[.., lo % G]
            float nv[4], av[4];
#pragma unroll
            for (int v = 0; v < 4; v++) {
                if (APPLY) { nv[v] = 0.f; av[v] = t.vD ? b.y[v] : 0.f; }
                else { nv[v] = (t.vD && lo < G) ? fmaf(ga1[2], b.u[v], fmaf(ga2[2], b.y[v], ga3[2])) : 0.f; av[v] = 0.f; }
            }
            // APPLY: p1 of this lane's four slots (row 20 of the staged tile, [slot][3]) and the d p1 partial sums over this lane's channels
            float pq[12], t3[12];
#pragma unroll
            for (int e = 0; e < 12; e++) { pq[e] = 0.f; t3[e] = 0.f; }
            if (APPLY) {
#pragma unroll
                for (int e4 = 0; e4 < 3; e4++) {
                    const float4 x = *reinterpret_cast<const float4*>(&T[20][12 * hi + 4 * e4]);
                    pq[4 * e4] = x.x; pq[4 * e4 + 1] = x.y; pq[4 * e4 + 2] = x.z; pq[4 * e4 + 3] = x.w;
                }
            }
            int lo_t = lo;
            if (TAB) asm volatile("" : "+v"(lo_t));                   // opaque per tile: the table's constant reads stay LDS reads inside the loop (not hoisted back into 40 registers)
#pragma unroll
            for (int ct = 0; ct < CT; ct++) {
                const int cc = 16 * ct + lo_t;
                const float sc_c = TAB ? ctab[cc] : sc[ct], sh_c = TAB ? ctab[C + cc] : sh[ct], k1_c = TAB ? ctab[2 * C + cc] : k1[ct],
                            k2_c = TAB ? ctab[3 * C + cc] : k2[ct], k3_c = APPLY ? ctab[4 * C + cc] : 0.f;
                const float wa0_c = TAB ? ctab[5 * C + hi * C + cc] : wa0[ct], wa1_c = TAB ? (G == 8 ? ctab[5 * C + (hi + 4) * C + cc] : 0.f) : wa1[ct];
                const float w3x = APPLY ? ctab[(5 + G) * C + cc] : 0.f, w3y = APPLY ? ctab[(6 + G) * C + cc] : 0.f, w3z = APPLY ? ctab[(7 + G) * C + cc] : 0.f;
                const float q = T[16 + (4 * hi) / K][16 * ct + lo], go = APPLY ? T[18 + (4 * hi) / K][16 * ct + lo] : 0.f;
                pt_f32x4 w = pt_vec4(T[4 * hi][16 * ct + lo] - q, T[4 * hi + 1][16 * ct + lo] - q, T[4 * hi + 2][16 * ct + lo] - q, T[4 * hi + 3][16 * ct + lo] - q);
                w = pt_mfma(b.p1x, pe.w[ct], w);
                pt_f32x4 gy = pt_mfma(da0, wa0_c, pt_vec4(0.f, 0.f, 0.f, 0.f));
                if (G == 8) gy = pt_mfma(da1, wa1_c, gy);
                float sq = 0.f;
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const float y = fmaf(w[v], sc_c, sh_c);
                    const float g1 = y > 0.f ? gy[v] : 0.f;
                    if (APPLY) {
                        const float dw = t.vD ? fmaf(k1_c, g1, fmaf(k2_c, w[v], k3_c)) : 0.f;
                        sq += dw;
                        const float dpe = fmaf(go, av[v], dw);
                        // d p1 partials (contraction over this lane's channels; the 16 lanes of the row are summed below) and d [W3C | b3C] (contraction
                        // over this lane's pairs; the lane rows and waves are summed at the end of the kernel): 7 v_fma per (slot, channel)
                        t3[3 * v] = fmaf(w3x, dpe, t3[3 * v]); t3[3 * v + 1] = fmaf(w3y, dpe, t3[3 * v + 1]); t3[3 * v + 2] = fmaf(w3z, dpe, t3[3 * v + 2]);
                        accw[ct][0] = fmaf(dpe, pq[3 * v], accw[ct][0]); accw[ct][1] = fmaf(dpe, pq[3 * v + 1], accw[ct][1]);
                        accw[ct][2] = fmaf(dpe, pq[3 * v + 2], accw[ct][2]); accw[ct][3] += dpe;
                    } else {
                        if (t.vD) { s1[ct] += g1; s2[ct] = fmaf(g1, fmaf(w[v], k1_c, k2_c), s2[ct]); }
                        accw[ct] = pt_mfma(nv[v], fmaxf(y, 0.f), accw[ct]);      // D[g][channel] += d w2[slot][g] w1[slot][channel]
                    }
                }
                if (APPLY) {
                    sq = pt_point_sum<K>(sq);
                    if (t.vD && (K == 16 ? hi == 0 : (hi & 1) == 0)) gxq[(pt_ix)t.iD * C + 16 * ct + lo] = -sq;
                }
            }
            if (APPLY) {
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const float r0 = pt_row_sum(t3[3 * v]), r1 = pt_row_sum(t3[3 * v + 1]), r2 = pt_row_sum(t3[3 * v + 2]);
                    if (t.vD && lo < 3) gp1[3 * (pt_ix)(t.pD + v) + lo] = lo == 0 ? r0 : (lo == 1 ? r1 : r2);
                }
            }
        },
        [&]() {
            if (APPLY) return;
            // BN_g's backward coefficients from the narrow backward pass's partial rows, in every workgroup (pt_fin_backward); workgroup 0 writes d gamma_g, d beta_g, d ba
            const float* fo = pt_fin_backward<PT_BLOCK, 2 * G>(fin_g, G, lds);
            const int gs[3] = {hi, hi + 4, lo};
#pragma unroll
            for (int t = 0; t < 3; t++)
                if (gs[t] < G) { ga1[t] = fo[gs[t]]; ga2[t] = fo[8 + gs[t]]; ga3[t] = fo[16 + gs[t]]; }
            __syncthreads();                                         // the scratch becomes the waves' tiles
        });
    // workgroup partial row
    __syncthreads();                                                 // every wave is done with its tile
    float (*red)[W] = reinterpret_cast<float (*)[W]>(lds);
    if (APPLY) {
        // accw[ct][d]: this lane's sum over its slots for channel 16 ct + lo; the four lane rows are added here; stored as torch lays out Linear(3, C):
        // weight [c][d], then the bias
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            float x[4];
#pragma unroll
            for (int v = 0; v < 4; v++) { x[v] = accw[ct][v] + pt_xor16(accw[ct][v]); x[v] += pt_xor32(x[v]); }
            if (hi == 0) {
#pragma unroll
                for (int v = 0; v < 3; v++) red[wave][3 * (16 * ct + lo) + v] = x[v];
                red[wave][3 * C + 16 * ct + lo] = x[3];
            }
        }
    } else {
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            float a1 = s1[ct] + pt_xor16(s1[ct]), a2 = s2[ct] + pt_xor16(s2[ct]);
            a1 += pt_xor32(a1); a2 += pt_xor32(a2);
            if (hi == 0) { red[wave][16 * ct + lo] = a1; red[wave][C + 16 * ct + lo] = a2; }
#pragma unroll
            for (int v = 0; v < 4; v++)
                if (4 * hi + v < G) red[wave][2 * C + (4 * hi + v) * C + 16 * ct + lo] = accw[ct][v];
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < W; t += PT_BLOCK) {
        float s = 0.f;
        for (int wv = 0; wv < PT_WPB; wv++) s += red[wv][t];
        partial[(size_t)blockIdx.x * W + t] = s;
    }
}

// ---- p chain backward (lane = pair): the sums BN_p's backward and Linear(3,3)'s gradient need -------------------------------------------
// partial row: T1[a] = sum d, T2[a] = sum d p0hat[a], T3[a][b] = sum d[a] p_r[b]   with d = (p1 > 0) d p1
__global__ __launch_bounds__(PT_NARROW_BLOCK) void pt_pchain_bwd_kernel(long long npairs, const float* __restrict__ p_r, const float* __restrict__ p0,
                                                                        const float* __restrict__ p1, const float* __restrict__ gp1, const float* __restrict__ cst,
                                                                        float* __restrict__ partial, const float* __restrict__ gp1b = nullptr)
{
    __shared__ float red[PT_NARROW_BLOCK / 64][15];
    float is[3], nm[3], acc[15];
#pragma unroll
    for (int t = 0; t < 3; t++) { is[t] = cst[PT_CST_P + 12 + t]; nm[t] = -cst[PT_CST_P + 8 + t] * is[t]; }
#pragma unroll
    for (int t = 0; t < 15; t++) acc[t] = 0.f;
    for (long long p = (long long)blockIdx.x * PT_NARROW_BLOCK + threadIdx.x; p < npairs; p += (long long)gridDim.x * PT_NARROW_BLOCK) {
        float d[3], r[3];
#pragma unroll
        for (int t = 0; t < 3; t++) { d[t] = p1[3 * p + t] > 0.f ? (gp1b ? gp1[3 * p + t] + gp1b[3 * p + t] : gp1[3 * p + t]) : 0.f; r[t] = p_r[3 * p + t]; }
#pragma unroll
        for (int a = 0; a < 3; a++) {
            acc[a] += d[a]; acc[3 + a] = fmaf(d[a], fmaf(p0[3 * p + a], is[a], nm[a]), acc[3 + a]);
#pragma unroll
            for (int b = 0; b < 3; b++) acc[6 + 3 * a + b] = fmaf(d[a], r[b], acc[6 + 3 * a + b]);
        }
    }
#pragma unroll
    for (int t = 0; t < 15; t++) { const float s = pt_wave_sum(acc[t]); if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][t] = s; }
    __syncthreads();
    if (threadIdx.x < 15) {
        float s = 0.f;
        for (int wv = 0; wv < PT_NARROW_BLOCK / 64; wv++) s += red[wv][threadIdx.x];
        partial[(size_t)blockIdx.x * 15 + threadIdx.x] = s;
    }
}

// one workgroup: gradients of Linear(3,3) and BN_p from the backward sums T and the forward sums (BatchNorm's backward is linear in d)
__device__ __forceinline__ void pt_pchain_epilogue_body(int nrows, const float* __restrict__ partial, long long rows, const float* __restrict__ cst,
                                                        const float* __restrict__ gamma_p, float* __restrict__ g_Wp, float* __restrict__ g_bp,
                                                        float* __restrict__ g_gamma_p, float* __restrict__ g_beta_p)
{
    __shared__ double red[64][16];
    __shared__ double T[16];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    double acc = 0.0;
    if (cl < 15) for (int r = sl; r < nrows; r += 64) acc += (double)partial[(size_t)r * 15 + cl];
    red[sl][cl] = acc;
    __syncthreads();
    if (sl == 0) { double s = 0.0; for (int j = 0; j < 64; j++) s += red[j][cl]; T[cl] = s; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        const double N = (double)rows, mean = (double)cst[PT_CST_P + 8 + a], invstd = (double)cst[PT_CST_P + 12 + a];
        const double A = (double)gamma_p[a] * invstd, m1 = T[a] / N, m2 = T[3 + a] / N;
        g_gamma_p[a] = (float)T[3 + a]; g_beta_p[a] = (float)T[a];
        const double sum_p0 = (double)cst[PT_FS_P + a];
        // d p0 = A (d - m1 - p0hat m2):   sum d p0 = A (T1 - N m1 - m2 sum p0hat),   sum d p0[a] p_r[b] = A (T3 - m1 R[b] - m2 Q[a][b])
        g_bp[a] = (float)(A * (T[a] - N * m1 - m2 * invstd * (sum_p0 - N * mean)));
        for (int b = 0; b < 3; b++) {
            const double R = (double)cst[PT_FS_P + 6 + b], X = (double)cst[PT_FS_P + 9 + 3 * a + b];
            g_Wp[3 * a + b] = (float)(A * (T[6 + 3 * a + b] - m1 * R - m2 * invstd * (X - mean * R)));
        }
    }
}

__global__ __launch_bounds__(PT_FIN_THREADS) void pt_pchain_epilogue_kernel(int nrows, const float* __restrict__ partial, long long rows, const float* __restrict__ cst,
                                                                            const float* __restrict__ gamma_p, float* __restrict__ g_Wp, float* __restrict__ g_bp,
                                                                            float* __restrict__ g_gamma_p, float* __restrict__ g_beta_p)
{
    pt_pchain_epilogue_body(nrows, partial, rows, cst, gamma_p, g_Wp, g_bp, g_gamma_p, g_beta_p);
}

// ---- target pass: d x_k[j] and d x_v[j] as gathers over the transposed neighbour table -----------------------------------------------
// C / 4 lanes own a target row (lane = 4 channels); the pairs that list the target are walked in ascending order; the chain of a pair is
// rebuilt from the target's own x_k row, x_q[i], p1 and d w2 with the fmaf order of the matrix instruction (same w bits, same ReLU mask).
template <int C, int K, int PP>
__global__ __launch_bounds__(256) void pt_target_kernel(unsigned n, const int* __restrict__ order, const int* __restrict__ inv_start, const int* __restrict__ inv_src,
                                                        const float* __restrict__ xq, const float* __restrict__ xk, const float* __restrict__ p1,
                                                        const float* __restrict__ cst, const float* __restrict__ bc, const float* __restrict__ W3C,
                                                        const float* __restrict__ b3C, const float* __restrict__ Wa, const float* __restrict__ gw2,
                                                        const float* __restrict__ a, const float* __restrict__ gout, float* __restrict__ gxk, float* __restrict__ gxv)
{
    constexpr int LPR = C / 4, G = C / 8, TPB = 256 / LPR;
    const int m = threadIdx.x % LPR, grp = threadIdx.x / LPR, c0 = 4 * m;
    // the per-channel constants (W3C columns, bias, BN_c forward and backward coefficients, Wa: 9 + G values per channel) in an LDS table, one 16-byte entry per
    // (quantity, channel quad), read where they are used: in registers they were 68 of the kernel's 166 (three waves per SIMD on a pass that waits for memory)
    constexpr int NQ = 9 + G;
    __shared__ float4 ctab[NQ][LPR];
    if ((int)threadIdx.x < LPR) {
        float q[NQ][4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int c = 4 * threadIdx.x + e;
            q[0][e] = W3C[3 * c]; q[1][e] = W3C[3 * c + 1]; q[2][e] = W3C[3 * c + 2]; q[3][e] = b3C[c];
            q[4][e] = cst[PT_CST_C + c]; q[5][e] = cst[PT_CST_C + 64 + c];
            q[6][e] = bc[PT_BC_C + c]; q[7][e] = bc[PT_BC_C + 64 + c]; q[8][e] = bc[PT_BC_C + 128 + c];
#pragma unroll
            for (int g = 0; g < G; g++) q[9 + g][e] = Wa[(size_t)g * C + c];
        }
#pragma unroll
        for (int k = 0; k < NQ; k++) ctab[k][threadIdx.x] = make_float4(q[k][0], q[k][1], q[k][2], q[k][3]);
    }
    __syncthreads();
    const unsigned ntrips = (n + TPB - 1) / TPB;
    for (unsigned v = blockIdx.x; v < 8u * cbl_xcd_per(ntrips); v += gridDim.x) {
        const unsigned tr = cbl_xcd_slot(v, ntrips) * TPB + grp;
        if (tr >= n) continue;
        const int j = order ? order[tr] : (int)tr;
        const int e0 = inv_start[tr], e1 = inv_start[tr + 1];
        const float4 kj = *reinterpret_cast<const float4*>(xk + (size_t)j * C + c0);
        const float kx[4] = {kj.x, kj.y, kj.z, kj.w};
        float ak[4] = {0.f, 0.f, 0.f, 0.f}, av[4] = {0.f, 0.f, 0.f, 0.f};
        // PP pairs per trip, their rows requested together; the ids of the NEXT PP pairs are requested before this trip's arithmetic
        // (a list's walk was one dependent id -> rows round trip per pair: 0.74 of the wave cycles parked on memory)
        int e = e0;
        unsigned pn[PP];
#pragma unroll
        for (int u = 0; u < PP; u++) pn[u] = e + u < e1 ? (unsigned)inv_src[e + u] : (u ? pn[0] : 0u);
        for (; e < e1; e += PP) {
            unsigned pp[PP];
#pragma unroll
            for (int u = 0; u < PP; u++) pp[u] = pn[u];
            float4 q4[PP], g4[PP], a4[PP], d0[PP], d1[PP]; float b0[PP], b1[PP], b2[PP];
#pragma unroll
            for (int u = 0; u < PP; u++) {
                const unsigned p = pp[u], i = p / (unsigned)K;
                q4[u] = *reinterpret_cast<const float4*>(xq + (size_t)i * C + c0);
                g4[u] = *reinterpret_cast<const float4*>(gout + (size_t)i * C + c0);
                a4[u] = *reinterpret_cast<const float4*>(a + (size_t)p * G + (c0 % G));
                b0[u] = p1[3 * (size_t)p]; b1[u] = p1[3 * (size_t)p + 1]; b2[u] = p1[3 * (size_t)p + 2];
                d0[u] = *reinterpret_cast<const float4*>(gw2 + (size_t)p * G);
                d1[u] = G == 8 ? *reinterpret_cast<const float4*>(gw2 + (size_t)p * G + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < PP; u++) pn[u] = e + PP + u < e1 ? (unsigned)inv_src[e + PP + u] : (u ? pn[0] : 0u);
            int mo = m;
            asm volatile("" : "+v"(mo));                              // opaque per trip: the table reads stay LDS reads inside the loop
            float w0[4], w1[4], w2c[4], wbias[4], sc[4], sh[4], k1[4], k2[4], k3[4], wa[G][4];
            {
                auto get = [&](int k, float (&o)[4]) { const float4 t = ctab[k][mo]; o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w; };
                get(0, w0); get(1, w1); get(2, w2c); get(3, wbias); get(4, sc); get(5, sh); get(6, k1); get(7, k2); get(8, k3);
#pragma unroll
                for (int g = 0; g < G; g++) get(9 + g, wa[g]);
            }
#pragma unroll
            for (int u = 0; u < PP; u++) {
                const float live = (e + u < e1) ? 1.f : 0.f;          // a list's last trip repeats a pair with weight 0 where it runs past the end
                const float qx[4] = {q4[u].x, q4[u].y, q4[u].z, q4[u].w}, gx[4] = {g4[u].x, g4[u].y, g4[u].z, g4[u].w}, ax[4] = {a4[u].x, a4[u].y, a4[u].z, a4[u].w};
                const float dv[8] = {d0[u].x, d0[u].y, d0[u].z, d0[u].w, d1[u].x, d1[u].y, d1[u].z, d1[u].w};
#pragma unroll
                for (int e4 = 0; e4 < 4; e4++) {
                    float w = kx[e4] - qx[e4];
                    w = fmaf(w0[e4], b0[u], w); w = fmaf(w1[e4], b1[u], w); w = fmaf(w2c[e4], b2[u], w); w = fmaf(wbias[e4], 1.f, w);
                    float gy = 0.f;
#pragma unroll
                    for (int g = 0; g < G; g++) gy = fmaf(dv[g], wa[g][e4], gy);
                    const float g1 = fmaf(w, sc[e4], sh[e4]) > 0.f ? gy : 0.f;
                    ak[e4] = fmaf(live, fmaf(k1[e4], g1, fmaf(k2[e4], w, k3[e4])), ak[e4]);
                    av[e4] = fmaf(live * gx[e4], ax[e4], av[e4]);
                }
            }
        }
        *reinterpret_cast<float4*>(gxk + (size_t)j * C + c0) = make_float4(ak[0], ak[1], ak[2], ak[3]);
        *reinterpret_cast<float4*>(gxv + (size_t)j * C + c0) = make_float4(av[0], av[1], av[2], av[3]);
    }
}

// one wave: D = A . B + C on the matrix instruction, and the same tile as the k-ordered fmaf chain the target pass uses (self-test of the
// assumption that both produce the same bits, on which the agreement of the passes' ReLU masks rests)
__global__ __launch_bounds__(64) void pt_chain_selftest_kernel(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ Cin,
                                                               float* __restrict__ d_mfma, float* __restrict__ d_fma)
{
    const int lane = threadIdx.x, lo = lane & 15, hi = lane >> 4;
    pt_f32x4 c = pt_vec4(Cin[(4 * hi) * 16 + lo], Cin[(4 * hi + 1) * 16 + lo], Cin[(4 * hi + 2) * 16 + lo], Cin[(4 * hi + 3) * 16 + lo]);
    const pt_f32x4 d = pt_mfma(A[lo * 4 + hi], B[hi * 16 + lo], c);                // A (16 x 4) row-major, B (4 x 16) row-major, C / D (16 x 16) row-major
    for (int v = 0; v < 4; v++) {
        const int row = 4 * hi + v;
        float t = Cin[row * 16 + lo];
        for (int k = 0; k < 4; k++) t = fmaf(A[row * 4 + k], B[k * 16 + lo], t);
        d_mfma[row * 16 + lo] = d[v];
        d_fma[row * 16 + lo] = t;
    }
}

// the backward pass's last launch: the parameter gradients' column sums (pt_sum_rows_kernel's blocks) and, in the block behind them, the p chain's epilogue
__global__ __launch_bounds__(PT_FIN_THREADS) void pt_bwd_tail_kernel(PtSumSegs segs, int sum_blocks, int nrows, const float* __restrict__ partial, long long rows,
                                                                     const float* __restrict__ cst, const float* __restrict__ gamma_p, float* __restrict__ g_Wp,
                                                                     float* __restrict__ g_bp, float* __restrict__ g_gamma_p, float* __restrict__ g_beta_p)
{
    if ((int)blockIdx.x < sum_blocks) pt_sum_rows_body(segs);
    else pt_pchain_epilogue_body(nrows, partial, rows, cst, gamma_p, g_Wp, g_bp, g_gamma_p, g_beta_p);
}

unsigned pt_tile_grid(long long ntiles, int max_rows = PT_MAX_ROWS)
{
    const long long trips = (ntiles + PT_WPB - 1) / PT_WPB;
    long long g = trips < max_rows ? trips : max_rows;
    g = (g + 7) & ~7ll;
    return (unsigned)(g < 8 ? 8 : g);
}
// workgroups (= partial rows) of the lane-per-pair passes (p chain, narrow backward): they stream (n, K, 3 | G) tensors with one dependent
// round trip per iteration, so they want more waves in flight than the tile passes do
constexpr int PT_NARROW_MAX_ROWS = 2048;
int pt_narrow_rows()
{
    static const int rows = [] {
        const char* e = getenv("CBL_PT_NARROW_ROWS");               // measurement knob (tools/gpu_r05_call2.sh); default below
        int v = e ? atoi(e) : 512;
        return v < 1 ? 1 : (v > PT_NARROW_MAX_ROWS ? PT_NARROW_MAX_ROWS : v);
    }();
    return rows;
}
unsigned pt_pair_grid(long long npairs)
{
    long long g = (npairs + PT_NARROW_BLOCK - 1) / PT_NARROW_BLOCK;
    if (g > pt_narrow_rows()) g = pt_narrow_rows();
    return (unsigned)(g < 1 ? 1 : g);
}

// workspace (floats): partial rows of every pass + the backward's narrow tensors
struct PtWs { float *part_a, *part_b, *part_c, *part_d, *bc, *glogit, *pre, *gw2, *gp1; size_t floats; };
PtWs pt_workspace(float* base, int n, int K, int C)
{
    const size_t np = (size_t)n * K, G = C / 8;
    PtWs w; size_t o = 0;
    auto take = [&](size_t cnt) { float* p = base ? base + o : nullptr; o += (cnt + 63) & ~(size_t)63; return p; };
    w.part_a = take((size_t)PT_MAX_ROWS * (2 * C + G * C));      // wstats, w2, reduce
    w.part_b = take((size_t)PT_MAX_ROWS * (4 * C));              // pchain (PT_NARROW_MAX_ROWS x 18 fits: static_assert below), apply
    w.part_c = take((size_t)PT_NARROW_MAX_ROWS * (3 * G + G * G));   // narrow backward
    w.part_d = take((size_t)PT_NARROW_MAX_ROWS * 16);            // pchain backward
    w.bc = take(PT_BC_FLOATS);
    w.glogit = take(np * G); w.pre = take(np * G); w.gw2 = take(np * G); w.gp1 = take(np * 3);
    w.floats = o;
    return w;
}

static_assert(PT_NARROW_MAX_ROWS * 18 <= PT_MAX_ROWS * 4 * 32, "pchain's partial rows share part_b with the apply pass");
static_assert((PT_BLOCK * 18 + 2 * (PT_BLOCK + 32)) <= PT_WPB * PT_TROWS * PT_ROWF, "the in-consumer finalize's scratch (rows of a chunk + the doubles) lies over the waves' tiles");
bool pt_shape_ok(int n, int K, int C) { return n >= 1 && (K == 8 || K == 16) && (C == 32 || C == 64) && (long long)n * K < (1ll << 28); }   // 32-bit element indices (pt_ix): n K 8 and n C below 2^31

}  // namespace

CBL_EXPORT size_t cbl_pt_layer_workspace_bytes(int n, int K, int C)
{
    if (!pt_shape_ok(n, K, C)) return 0;
    return pt_workspace(nullptr, n, K, C).floats * sizeof(float);
}
CBL_EXPORT int cbl_pt_layer_consts_floats(void) { return PT_CST_FLOATS; }

CBL_EXPORT int cbl_pt_layer_selftest_chain(const float* A, const float* B, const float* C, float* d_mfma, float* d_fma, void* stream)
{
    hipLaunchKernelGGL(pt_chain_selftest_kernel, dim3(1), dim3(64), 0, cbl_stream(stream), A, B, C, d_mfma, d_fma);
    return cbl_status();
}

#define PT_DISPATCH(CALL)                                     \
    if (C == 64 && K == 16) { CALL(64, 16); }                 \
    else if (C == 64 && K == 8) { CALL(64, 8); }              \
    else if (C == 32 && K == 16) { CALL(32, 16); }            \
    else { CALL(32, 8); }

CBL_EXPORT int cbl_pt_layer_forward(int n, int K, int C, const float* xyz, const float* x_q, const float* x_k, const float* x_v, const int* idx, const int* order,
                                    const float* Wp, const float* bp, const float* gamma_p, const float* beta_p, const float* W3C, const float* b3C,
                                    const float* gamma_c, const float* beta_c, const float* Wa, const float* ba, const float* gamma_g, const float* beta_g,
                                    const float* Wb, const float* bb, const float* eps3, const float* momentum3, float* const* running_mean3,
                                    float* const* running_var3, long long* const* num_batches3, float* p_r, float* p0, float* p1, float* w2, float* a, float* out,
                                    float* consts, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!pt_shape_ok(n, K, C)) return CBL_ERR_UNSUPPORTED;
    if (!cbl_host_aligned16(x_q) || !cbl_host_aligned16(x_k) || !cbl_host_aligned16(x_v) || !cbl_host_aligned16(w2) || !cbl_host_aligned16(a) ||
        !cbl_host_aligned16(idx) || !cbl_host_aligned16(consts) || !cbl_host_aligned16(Wa))
        return CBL_ERR_BAD_ARG;
    if (workspace_bytes < cbl_pt_layer_workspace_bytes(n, K, C)) return CBL_ERR_WORKSPACE;
    const PtWs ws = pt_workspace(static_cast<float*>(workspace), n, K, C);
    hipStream_t st = cbl_stream(stream);
    const long long np = (long long)n * K;
    const int G = C / 8;
    const unsigned gp = pt_pair_grid(np), gt = pt_tile_grid((np + 15) / 16);
    // the passes that hold one workgroup per CU (reduce, apply: 148 - 189 registers; w2 at C = 64: 139): one per CU in the launch, so that the pass's start-up (constants,
    // pipeline fill, in-consumer finalize) is paid once — the reduce pass 61.7 -> 53.4 us at (40960, 16, 64)
    const unsigned gt1 = pt_tile_grid((np + 15) / 16, PT_ONE_PER_CU), gw = C == 64 ? gt1 : gt;
    float* rm[3] = {nullptr, nullptr, nullptr}; float* rv[3] = {nullptr, nullptr, nullptr}; long long* nb[3] = {nullptr, nullptr, nullptr};
    for (int t = 0; t < 3; t++) { if (running_mean3) rm[t] = running_mean3[t]; if (running_var3) rv[t] = running_var3[t]; if (num_batches3) nb[t] = num_batches3[t]; }

    hipLaunchKernelGGL(pt_pchain_kernel, dim3(gp), dim3(PT_NARROW_BLOCK), 0, st, np, cbl_fastdiv_make((unsigned)K), xyz, idx, Wp, bp, p_r, p0, ws.part_b, (const float*)nullptr, (float*)nullptr);
    // BN_p's finalize runs in the prologue of the statistics pass, BN_g's in the prologue of the softmax + aggregation pass (pt_fin_forward); BN_c keeps its launch
    const PtFin fin_p = {ws.part_b, (int)gp, 18, np, gamma_p, beta_p, eps3[0], momentum3[0], rm[0], rv[0], nb[0], consts};
    const PtFin fin_g = {ws.part_a, (int)gw, 2 * G, np, gamma_g, beta_g, eps3[2], momentum3[2], rm[2], rv[2], nb[2], consts};
#define PT_WSTATS(CC, KK) hipLaunchKernelGGL((pt_wstats_kernel<CC, KK>), dim3(gt), dim3(PT_BLOCK), 0, st, n, order, x_q, x_k, idx, p0, fin_p, W3C, b3C, p1, ws.part_a)
    PT_DISPATCH(PT_WSTATS)
    hipLaunchKernelGGL(pt_bn_finalize_kernel, dim3(cbl_div_up(C, 16)), dim3(PT_FIN_THREADS), 0, st, (int)gt, 2 * C, ws.part_a, C, 0, C, np, gamma_c, beta_c, eps3[1],
                       momentum3[1], rm[1], rv[1], nb[1], consts + PT_CST_C, 64, 0, (float*)nullptr);
#define PT_W2(CC, KK) hipLaunchKernelGGL((pt_w2_kernel<CC, KK>), dim3(gw), dim3(PT_BLOCK), 0, st, n, order, x_q, x_k, idx, p1, consts, W3C, b3C, Wa, ba, w2, ws.part_a)
    PT_DISPATCH(PT_W2)
#define PT_AGG(CC, KK) hipLaunchKernelGGL((pt_agg_kernel<CC, KK, false>), dim3(gt), dim3(PT_BLOCK), 0, st, n, order, x_v, idx, p1, W3C, b3C, a, out, (const float*)nullptr, (float*)nullptr, (const float*)w2, Wb, bb, (const float*)consts, fin_g, (float*)nullptr)
    PT_DISPATCH(PT_AGG)
    return cbl_status();
}

namespace {
// evaluation mode: the three BatchNorms' constants from their running statistics (scale = gamma / sqrt(var + eps), shift = beta - mean scale)
__global__ __launch_bounds__(128) void pt_eval_consts_kernel(int C, int G, const float* __restrict__ gp, const float* __restrict__ bp, const float* __restrict__ gc,
                                                             const float* __restrict__ bc, const float* __restrict__ gg, const float* __restrict__ bg,
                                                             const float* __restrict__ mp, const float* __restrict__ vp, const float* __restrict__ mc,
                                                             const float* __restrict__ vc, const float* __restrict__ mg, const float* __restrict__ vg,
                                                             float ep, float ec, float eg, float* __restrict__ cst)
{
    const int t = threadIdx.x;
    auto put = [&](float* base, int stride, int c, float gamma, float beta, float mean, float var, float eps) {
        const float invstd = (float)(1.0 / sqrt((double)var + (double)eps)), scale = gamma * invstd;
        base[c] = scale; base[stride + c] = beta - mean * scale; base[2 * stride + c] = mean; base[3 * stride + c] = invstd;
    };
    if (t < 3) put(cst + PT_CST_P, 4, t, gp[t], bp[t], mp[t], vp[t], ep);
    if (t < C) put(cst + PT_CST_C, 64, t, gc[t], bc[t], mc[t], vc[t], ec);
    if (t < G) put(cst + PT_CST_G, 8, t, gg[t], bg[t], mg[t], vg[t], eg);
}
}  // namespace

/* evaluation mode of the same layer (model.eval(): BatchNorm1d normalises with its running statistics, blocks.py:38-40 under nn.Module.eval): no statistics
 * passes, no buffers touched — p chain, w2, softmax, aggregation; outputs as cbl_pt_layer_forward (w2 / a / p1 are scratch of the caller). */
CBL_EXPORT int cbl_pt_layer_forward_eval(int n, int K, int C, const float* xyz, const float* x_q, const float* x_k, const float* x_v, const int* idx, const int* order,
                                         const float* Wp, const float* bp, const float* gamma_p, const float* beta_p, const float* W3C, const float* b3C,
                                         const float* gamma_c, const float* beta_c, const float* Wa, const float* ba, const float* gamma_g, const float* beta_g,
                                         const float* Wb, const float* bb, const float* eps3, const float* const* running_mean3, const float* const* running_var3,
                                         float* p_r, float* p0, float* p1, float* w2, float* a, float* out, float* consts, void* workspace, size_t workspace_bytes,
                                         void* stream)
{
    if (!pt_shape_ok(n, K, C)) return CBL_ERR_UNSUPPORTED;
    if (!running_mean3 || !running_var3 || !eps3) return CBL_ERR_BAD_ARG;
    if (!cbl_host_aligned16(x_q) || !cbl_host_aligned16(x_k) || !cbl_host_aligned16(x_v) || !cbl_host_aligned16(w2) || !cbl_host_aligned16(a) ||
        !cbl_host_aligned16(idx) || !cbl_host_aligned16(consts) || !cbl_host_aligned16(Wa))
        return CBL_ERR_BAD_ARG;
    if (workspace_bytes < cbl_pt_layer_workspace_bytes(n, K, C)) return CBL_ERR_WORKSPACE;
    const PtWs ws = pt_workspace(static_cast<float*>(workspace), n, K, C);
    hipStream_t st = cbl_stream(stream);
    const long long np = (long long)n * K;
    const unsigned gp = pt_pair_grid(np), gt = pt_tile_grid((np + 15) / 16);
    // the passes that hold one workgroup per CU (reduce, apply: 148 - 189 registers; w2 at C = 64: 139): one per CU in the launch, so that the pass's start-up (constants,
    // pipeline fill, in-consumer finalize) is paid once — the reduce pass 61.7 -> 53.4 us at (40960, 16, 64)
    const unsigned gt1 = pt_tile_grid((np + 15) / 16, PT_ONE_PER_CU), gw = C == 64 ? gt1 : gt;
    const int G = C / 8;
    hipLaunchKernelGGL(pt_eval_consts_kernel, dim3(1), dim3(128), 0, st, C, G, gamma_p, beta_p, gamma_c, beta_c, gamma_g, beta_g, running_mean3[0], running_var3[0],
                       running_mean3[1], running_var3[1], running_mean3[2], running_var3[2], eps3[0], eps3[1], eps3[2], consts);
    hipLaunchKernelGGL(pt_pchain_kernel, dim3(gp), dim3(PT_NARROW_BLOCK), 0, st, np, cbl_fastdiv_make((unsigned)K), xyz, idx, Wp, bp, p_r, p0, ws.part_b,
                       (const float*)consts, p1);
    PT_DISPATCH(PT_W2)
    const PtFin fin_g = {nullptr, 0, 0, np, nullptr, nullptr, 0.f, 0.f, nullptr, nullptr, nullptr, consts};      // BN_g's constants are in consts already
    PT_DISPATCH(PT_AGG)
    return cbl_status();
}

CBL_EXPORT int cbl_pt_layer_backward(int n, int K, int C, const float* x_q, const float* x_k, const float* x_v, const int* idx, const int* order,
                                     const int* inv_start, const int* inv_src, const float* gamma_p, const float* W3C, const float* b3C, const float* gamma_c,
                                     const float* Wa, const float* gamma_g, const float* Wb, const float* p_r, const float* p0, const float* p1, const float* w2,
                                     const float* a, const float* consts, const float* grad_out, float* g_xq, float* g_xk, float* g_xv, float* g_Wp, float* g_bp,
                                     float* g_gamma_p, float* g_beta_p, float* g_W3C, float* g_b3C, float* g_gamma_c, float* g_beta_c, float* g_Wa, float* g_ba,
                                     float* g_gamma_g, float* g_beta_g, float* g_Wb, float* g_bb, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!pt_shape_ok(n, K, C)) return CBL_ERR_UNSUPPORTED;
    if (!cbl_host_aligned16(x_q) || !cbl_host_aligned16(x_k) || !cbl_host_aligned16(x_v) || !cbl_host_aligned16(w2) || !cbl_host_aligned16(a) ||
        !cbl_host_aligned16(idx) || !cbl_host_aligned16(grad_out) || !cbl_host_aligned16(g_xk) || !cbl_host_aligned16(g_xv))
        return CBL_ERR_BAD_ARG;
    if (workspace_bytes < cbl_pt_layer_workspace_bytes(n, K, C)) return CBL_ERR_WORKSPACE;
    const PtWs ws = pt_workspace(static_cast<float*>(workspace), n, K, C);
    hipStream_t st = cbl_stream(stream);
    const long long np = (long long)n * K;
    const int G = C / 8, WN = 3 * G + G * G;
    const unsigned gp = pt_pair_grid(np), gt = pt_tile_grid((np + 15) / 16);
    // the passes that hold one workgroup per CU (reduce, apply: 148 - 189 registers; w2 at C = 64: 139): one per CU in the launch, so that the pass's start-up (constants,
    // pipeline fill, in-consumer finalize) is paid once — the reduce pass 61.7 -> 53.4 us at (40960, 16, 64)
    const unsigned gt1 = pt_tile_grid((np + 15) / 16, PT_ONE_PER_CU), gw = C == 64 ? gt1 : gt;

    const PtFin no_fin = {nullptr, 0, 0, np, nullptr, nullptr, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr};
    const unsigned gab = gt1;                                        // 155 registers with the narrow backward inside: one workgroup per CU (against two per CU launched: 34.7 -> 31.9 us, and the reduce pass's prologue reads half the rows)
#define PT_AGGB(CC, KK) hipLaunchKernelGGL((pt_agg_kernel<CC, KK, true>), dim3(gab), dim3(PT_BLOCK), 0, st, n, order, x_v, idx, p1, W3C, b3C, const_cast<float*>(a), (float*)nullptr, grad_out, ws.pre, w2, Wb, (const float*)nullptr, consts, no_fin, ws.part_c)
    PT_DISPATCH(PT_AGGB)                                             // d logits and the narrow backward behind them: writes `pre`, partial rows in part_c
    // BN_g's backward finalize runs in the prologue of the reduce pass (pt_fin_backward)
    const PtFinBwd fin_gb = {ws.part_c, (int)gab, WN, np, gamma_g, consts + PT_CST_G, 8, consts + PT_FS_G, g_gamma_g, g_beta_g, g_ba};
    const PtFinBwd no_finb = {nullptr, 0, 0, np, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr};
#define PT_REDUCE(CC, KK) hipLaunchKernelGGL((pt_w2_bwd_kernel<CC, KK, false>), dim3(gt1), dim3(PT_BLOCK), 0, st, n, order, x_q, x_k, idx, p1, consts, ws.bc, W3C, b3C, Wa, w2, ws.pre, ws.gw2, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, ws.part_a, fin_gb)
    PT_DISPATCH(PT_REDUCE)
    hipLaunchKernelGGL(pt_bn_bwd_finalize_kernel, dim3(cbl_div_up(C, 16)), dim3(PT_FIN_THREADS), 0, st, (int)gt1, 2 * C + G * C, 0, C, C, np, ws.part_a, gamma_c, consts + PT_CST_C, 64,
                       ws.bc + PT_BC_C, 64, g_gamma_c, g_beta_c, (const float*)nullptr, (float*)nullptr);
#define PT_APPLY(CC, KK) hipLaunchKernelGGL((pt_w2_bwd_kernel<CC, KK, true>), dim3(gt1), dim3(PT_BLOCK), 0, st, n, order, x_q, x_k, idx, p1, consts, ws.bc, W3C, b3C, Wa, w2, ws.pre, ws.gw2, a, grad_out, g_xq, ws.gp1, ws.part_b, no_finb)
    PT_DISPATCH(PT_APPLY)
    hipLaunchKernelGGL(pt_pchain_bwd_kernel, dim3(gp), dim3(PT_NARROW_BLOCK), 0, st, np, p_r, p0, p1, ws.gp1, consts, ws.part_d, (const float*)nullptr);
    {
        const unsigned tg = cbl_round_up8(cbl_grid_for(((long long)n + (256 / (C / 4)) - 1) / (256 / (C / 4)), 1, 2048));
#define PT_TARGET(CC, KK) hipLaunchKernelGGL((pt_target_kernel<CC, KK, PT_TARGET_PAIRS(CC)>), dim3(tg), dim3(256), 0, st, (unsigned)n, order, inv_start, inv_src, x_q, x_k, p1, consts, ws.bc, W3C, b3C, Wa, ws.gw2, a, grad_out, g_xk, g_xv)
        PT_DISPATCH(PT_TARGET)
    }
    PtSumSegs segs;
    segs.n = 5;
    segs.s[0] = PtSumSeg{ws.part_a, g_Wa, (int)gt1, 2 * C + G * C, 2 * C, G * C};
    segs.s[1] = PtSumSeg{ws.part_b, g_W3C, (int)gt1, 4 * C, 0, 3 * C};
    segs.s[2] = PtSumSeg{ws.part_b, g_b3C, (int)gt1, 4 * C, 3 * C, C};
    segs.s[3] = PtSumSeg{ws.part_c, g_Wb, (int)gab, WN, 2 * G, G * G};
    segs.s[4] = PtSumSeg{ws.part_c, g_bb, (int)gab, WN, 2 * G + G * G, G};
    unsigned nblk = 0;
    for (int q = 0; q < segs.n; q++) nblk += (unsigned)((segs.s[q].count + 15) / 16);
    // + the p chain's epilogue as one more block of the same launch
    hipLaunchKernelGGL(pt_bwd_tail_kernel, dim3(nblk + 1), dim3(PT_FIN_THREADS), 0, st, segs, (int)nblk, (int)gp, ws.part_d, np, consts, gamma_p, g_Wp, g_bp, g_gamma_p, g_beta_p);
    return cbl_status();
}


#ifndef CBL_HOST_WAVE_EMULATION     // (the CPU build of this file for tests/test_pt_layer_host.py stops here: the section below calls attention.hip's entry points and the runtime)
// =====================================================================================================================================
// The WIDE stages (C = 128 / 256 / 512, G = C / 8 = 16 / 32 / 64, K = 16; n = 2560 / 640 / 160 points of a 40960-point scene): 13 of the network's 18
// Point Transformer blocks.  Their C-wide work already runs as six fused kernels (csrc/attention.hip: statistics, w2, aggregation, and the three
// backward passes — pair values recomputed, nothing (n, K, C) stored), but everything around them was issued op by op through autograd: ~42 launches
// per layer and pass pair behind the projections, most of them at their launch floor (profiles/r05_wide_layer_kernel_stats.csv).  These two entries are the whole layer behind its q / k / v
// projections as ONE call each way: the p chain, BN_p / BN_g statistics and their finalizes (the kernels of the full-resolution layer above: they do not
// depend on C), the six attention.hip kernels through their C entries, and four small kernels of this file for the narrow (n, K, G) tensors at any
// G <= 64 — ~9 launches forward, ~14 backward, no allocator, no autograd engine, no gradient-accumulation adds in between.
// =====================================================================================================================================
namespace {

constexpr int PW_CST_G = 352;                       // [4][64]  scale, shift, mean, invstd of BN_g (G <= 64); BN_p stays at PT_CST_P, the p sums at PT_FS_P
constexpr int PW_FS_G = 608;                        // raw sums of w2 [64]
constexpr int PW_CST_FLOATS = 672;
constexpr int PW_ROWS = 256;                        // partial rows (= workgroups) of the statistics kernel below
constexpr int PW_NROWS = 1024;                      // ... of the narrow backward: its per-batch work is a chain of LDS round trips, so short chains on many workgroups

// p1 = ReLU(BN_p(p0)) on (n K, 3): the attention.hip kernels take it materialised
__global__ __launch_bounds__(256) void pw_p1_kernel(long long total, const float* __restrict__ p0, const float* __restrict__ cst, float* __restrict__ p1)
{
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int t = (int)(e % 3);
        p1[e] = fmaxf(fmaf(p0[e], cst[PT_CST_P + t], cst[PT_CST_P + 4 + t]), 0.f);
    }
}

// BN_g statistics of w2 (n K, G): partial row = sum [G] | sum of squares [G]; thread (slice, g) walks the workgroup's pairs
__global__ __launch_bounds__(256) void pw_gstats_kernel(long long npairs, int G, const float* __restrict__ w2, float* __restrict__ partial)
{
    __shared__ float red[2][256];
    const int g = threadIdx.x % G, sl = threadIdx.x / G, nsl = 256 / G;
    const long long per = (npairs + gridDim.x - 1) / gridDim.x, p0 = (long long)blockIdx.x * per, p1 = min(npairs, p0 + per);
    float a = 0.f, b = 0.f;
    for (long long p = p0 + sl; p < p1; p += nsl) { const float x = w2[p * G + g]; a += x; b = fmaf(x, x, b); }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
    __syncthreads();
    if (threadIdx.x < G) {
        float sa = 0.f, sb = 0.f;
        for (int q = 0; q < nsl; q++) { sa += red[0][q * G + threadIdx.x]; sb += red[1][q * G + threadIdx.x]; }
        partial[(size_t)blockIdx.x * (2 * G) + threadIdx.x] = sa; partial[(size_t)blockIdx.x * (2 * G) + G + threadIdx.x] = sb;
    }
}

// logits = Linear(G, G)(ReLU(BN_g(w2))): a batch of 256 / G pairs per trip, thread (pair of the batch, output o); Wb in LDS (rows padded by one: threads of
// consecutive o read consecutive rows)
template <int G>
__global__ __launch_bounds__(256) void pw_logits_kernel(long long npairs, const float* __restrict__ w2, const float* __restrict__ cst, const float* __restrict__ Wb,
                                                        const float* __restrict__ bb, float* __restrict__ logits)
{
    extern __shared__ float lds[];
    float* wbs = lds;                                               // [G][G + 1]
    float* w3s = lds + G * (G + 1);                                 // [256 / G][G]
    for (int e = threadIdx.x; e < G * G; e += 256) wbs[(e / G) * (G + 1) + e % G] = Wb[e];
    const int o = threadIdx.x % G, pl = threadIdx.x / G, PB = 256 / G;
    const float sc = cst[PW_CST_G + o], sh = cst[PW_CST_G + 64 + o], bias = bb[o];
    const long long nb = (npairs + PB - 1) / PB;
    for (long long bt = blockIdx.x; bt < nb; bt += gridDim.x) {
        const long long p = bt * PB + pl;
        __syncthreads();                                            // the previous batch is consumed (first trip: Wb is in place)
        w3s[pl * G + o] = p < npairs ? fmaxf(fmaf(w2[p * G + o], sc, sh), 0.f) : 0.f;
        __syncthreads();
        float acc = bias;
#pragma unroll 16
        for (int g = 0; g < G; g++) acc = fmaf(wbs[o * (G + 1) + g], w3s[pl * G + g], acc);
        if (p < npairs) logits[p * G + o] = acc;
    }
}

// backward of Linear(G, G), ReLU and (its sums only) BN_g: pre = (y > 0) Wb^T d logits, written; partial row = S1 [G] | S2 [G] | d Wb [G][G] | d bb [G]
// (the layout of pt_narrow_bwd_kernel, so the finalize / sum kernels above serve both)
template <int G>
__global__ __launch_bounds__(256) void pw_narrow_bwd_kernel(long long npairs, const float* __restrict__ w2, const float* __restrict__ cst, const float* __restrict__ Wb,
                                                            const float* __restrict__ glogit, float* __restrict__ pre, float* __restrict__ partial)
{
    extern __shared__ float lds[];
    constexpr int PB = 256 / G, NE = G * G / 256;                   // pairs per batch; d Wb entries per thread (1, 4, 16)
    float* wbs = lds;                                               // [G][G]      (thread g reads column g: consecutive)
    float* gls = wbs + G * G;                                       // [PB][G]  d logits of the batch
    float* w3s = gls + 256;                                         // [PB][G]  ReLU(BN_g(w2))
    float* prs = w3s + 256;                                         // [PB][G]  pre
    float* pxs = prs + 256;                                         // [PB][G]  pre * xhat
    for (int e = threadIdx.x; e < G * G; e += 256) wbs[e] = Wb[e];
    const int g = threadIdx.x % G, pl = threadIdx.x / G;
    const float sc = cst[PW_CST_G + g], sh = cst[PW_CST_G + 64 + g], is = cst[PW_CST_G + 192 + g], nm = -cst[PW_CST_G + 128 + g] * is;
    float acc[NE], s1 = 0.f, s2 = 0.f, sb = 0.f;
#pragma unroll
    for (int q = 0; q < NE; q++) acc[q] = 0.f;
    const long long per = (npairs + gridDim.x - 1) / gridDim.x, b0 = (long long)blockIdx.x * per, b1 = min(npairs, b0 + per);
    for (long long base = b0; base < b1; base += PB) {
        const long long p = base + pl;
        const bool live = p < b1;
        __syncthreads();
        const float x = live ? w2[p * G + g] : 0.f, y = fmaf(x, sc, sh);
        gls[pl * G + g] = live ? glogit[p * G + g] : 0.f;
        w3s[pl * G + g] = live ? fmaxf(y, 0.f) : 0.f;
        __syncthreads();
        float sacc = 0.f;
#pragma unroll 16
        for (int o = 0; o < G; o++) sacc = fmaf(wbs[o * G + g], gls[pl * G + o], sacc);
        const float d = (live && y > 0.f) ? sacc : 0.f;
        if (live) pre[p * G + g] = d;
        prs[pl * G + g] = d; pxs[pl * G + g] = d * fmaf(x, is, nm);
        __syncthreads();
        if (threadIdx.x < G) {                                      // thread g: the batch's column sums (fixed order)
#pragma unroll
            for (int q = 0; q < PB; q++) { s1 += prs[q * G + g]; s2 += pxs[q * G + g]; sb += gls[q * G + g]; }
        }
#pragma unroll
        for (int q = 0; q < NE; q++) {
            const int e = threadIdx.x + 256 * q, eo = e / G, eg = e % G;
            float t = acc[q];
#pragma unroll
            for (int r = 0; r < PB; r++) t = fmaf(gls[r * G + eo], w3s[r * G + eg], t);
            acc[q] = t;
        }
    }
    float* row = partial + (size_t)blockIdx.x * (3 * G + G * G);
    if (threadIdx.x < G) { row[threadIdx.x] = s1; row[G + threadIdx.x] = s2; row[2 * G + G * G + threadIdx.x] = sb; }
#pragma unroll
    for (int q = 0; q < NE; q++) row[2 * G + threadIdx.x + 256 * q] = acc[q];
}

// d w2 = A1 pre + A2 w2 + A3  (BatchNorm backward with the batch sums folded into the three per-channel coefficients)
__global__ __launch_bounds__(256) void pw_bn_apply_kernel(long long total, int G, const float* __restrict__ pre, const float* __restrict__ w2, const float* __restrict__ bc,
                                                          float* __restrict__ gw2)
{
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int g = (int)(e % G);
        gw2[e] = fmaf(bc[g], pre[e], fmaf(bc[64 + g], w2[e], bc[128 + g]));
    }
}

struct PwWs { float *part_p, *part_g, *part_n, *part_d, *bc_g, *logits, *glogit, *pre, *gw2, *gp1a, *gp1b, *w3c2, *b3c2, *gba2; void* attn; size_t attn_bytes; size_t bytes; };
PwWs pw_workspace(char* base, int n, int K, int C)
{
    const size_t np = (size_t)n * K, G = C / 8;
    PwWs w; size_t o = 0;
    auto takeb = [&](size_t bytes) { char* p = base ? base + o : nullptr; o += (bytes + 255) & ~(size_t)255; return p; };
    auto take = [&](size_t cnt) { return reinterpret_cast<float*>(takeb(cnt * sizeof(float))); };
    w.part_p = take((size_t)PT_NARROW_MAX_ROWS * 18);
    w.part_g = take((size_t)PW_ROWS * 2 * G);
    w.part_n = take((size_t)PW_NROWS * (3 * G + G * G));
    w.part_d = take((size_t)PT_NARROW_MAX_ROWS * 16);
    w.bc_g = take(192);
    w.logits = take(np * G); w.glogit = take(np * G); w.pre = take(np * G); w.gw2 = take(np * G);
    w.gp1a = take(np * 3); w.gp1b = take(np * 3);
    w.w3c2 = take(2 * 3 * (size_t)C); w.b3c2 = take(2 * (size_t)C); w.gba2 = take(2 * G);
    w.attn_bytes = cbl_attn_workspace_bytes(C, (int)G);
    w.attn = takeb(w.attn_bytes);
    w.bytes = o;
    return w;
}
bool pw_shape_ok(int n, int K, int C) { return n >= 1 && K >= 1 && K <= 64 && (C == 128 || C == 256 || C == 512) && (long long)n * K < (1ll << 28); }

}  // namespace

CBL_EXPORT size_t cbl_pt_layer_wide_workspace_bytes(int n, int K, int C) { return pw_shape_ok(n, K, C) ? pw_workspace(nullptr, n, K, C).bytes : 0; }
CBL_EXPORT int cbl_pt_layer_wide_consts_floats(void) { return PW_CST_FLOATS; }

CBL_EXPORT int cbl_pt_layer_wide_forward(int n, int K, int C, const float* xyz, const float* x_q, const float* x_k, const float* x_v, const int* idx,
                                         const float* Wp, const float* bp, const float* gamma_p, const float* beta_p, const float* W3C, const float* b3C,
                                         const float* gamma_c, const float* beta_c, const float* Wa, const float* ba, const float* gamma_g, const float* beta_g,
                                         const float* Wb, const float* bb, const float* eps3, const float* momentum3, float* const* running_mean3,
                                         float* const* running_var3, long long* const* num_batches3, float* p_r, float* p0, float* p1, float* w2, float* a, float* out,
                                         float* consts, float* bnc_stats, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!pw_shape_ok(n, K, C)) return CBL_ERR_UNSUPPORTED;
    if (!eps3 || !momentum3 || !consts || !bnc_stats || !workspace) return CBL_ERR_BAD_ARG;
    if (workspace_bytes < cbl_pt_layer_wide_workspace_bytes(n, K, C)) return CBL_ERR_WORKSPACE;
    const PwWs ws = pw_workspace(static_cast<char*>(workspace), n, K, C);
    hipStream_t st = cbl_stream(stream);
    const long long np = (long long)n * K;
    const int G = C / 8;
    const unsigned gp = pt_pair_grid(np);
    float* rm[3] = {nullptr, nullptr, nullptr}; float* rv[3] = {nullptr, nullptr, nullptr}; long long* nb[3] = {nullptr, nullptr, nullptr};
    for (int t = 0; t < 3; t++) { if (running_mean3) rm[t] = running_mean3[t]; if (running_var3) rv[t] = running_var3[t]; if (num_batches3) nb[t] = num_batches3[t]; }
    int rc;
    // p chain + BN_p, then p1 materialised for the attention kernels
    hipLaunchKernelGGL(pt_pchain_kernel, dim3(gp), dim3(PT_NARROW_BLOCK), 0, st, np, cbl_fastdiv_make((unsigned)K), xyz, idx, Wp, bp, p_r, p0, ws.part_p, (const float*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL(pt_bn_finalize_kernel, dim3(2), dim3(PT_FIN_THREADS), 0, st, (int)gp, 18, ws.part_p, 3, 0, 3, np, gamma_p, beta_p, eps3[0], momentum3[0],
                       rm[0], rv[0], nb[0], consts + PT_CST_P, 4, 18, consts + PT_FS_P);
    hipLaunchKernelGGL(pw_p1_kernel, dim3(cbl_grid_for(3 * np, 256, 1024)), dim3(256), 0, st, 3 * np, p0, consts, p1);
    // w2 = Wa ReLU(BN_c(x_k[j] - x_q[i] + pe)) + ba: statistics pass, finalize (running statistics), product — attention.hip
    if ((rc = cbl_attn_w2_forward(n, K, C, G, x_q, x_k, idx, p1, W3C, b3C, gamma_c, beta_c, eps3[1], momentum3[1], rm[1], rv[1], nb[1], 1, Wa, ba,
                                  bnc_stats, bnc_stats + C, w2, ws.attn, ws.attn_bytes, stream))) return rc;
    // BN_g + ReLU + Linear(G, G), the softmax inside the aggregation kernel
    const unsigned gg = (unsigned)(np < PW_ROWS ? (np > 0 ? np : 1) : PW_ROWS);
    hipLaunchKernelGGL(pw_gstats_kernel, dim3(gg), dim3(256), 0, st, np, G, w2, ws.part_g);
    hipLaunchKernelGGL(pt_bn_finalize_kernel, dim3(cbl_div_up(G, 16)), dim3(PT_FIN_THREADS), 0, st, (int)gg, 2 * G, ws.part_g, G, 0, G, np, gamma_g, beta_g, eps3[2],
                       momentum3[2], rm[2], rv[2], nb[2], consts + PW_CST_G, 64, G, consts + PW_FS_G);
#define PW_LOGITS(GG) hipLaunchKernelGGL(pw_logits_kernel<GG>, dim3(cbl_grid_for(np * G, 256, 2048)), dim3(256), sizeof(float) * (G * (G + 1) + 256), st, np, w2, consts, Wb, bb, ws.logits)
    if (G == 16) { PW_LOGITS(16); } else if (G == 32) { PW_LOGITS(32); } else { PW_LOGITS(64); }
    if ((rc = cbl_attn_agg_softmax_forward(n, K, C, G, x_v, idx, p1, W3C, b3C, ws.logits, a, out, stream))) return rc;
    return cbl_status();
}

CBL_EXPORT int cbl_pt_layer_wide_backward(int n, int K, int C, const float* x_q, const float* x_k, const float* x_v, const int* idx, const float* gamma_p,
                                          const float* W3C, const float* b3C, const float* gamma_c, const float* beta_c, const float* Wa, const float* gamma_g,
                                          const float* Wb, const float* p_r, const float* p0, const float* p1, const float* w2, const float* a, const float* consts,
                                          const float* bnc_stats, const float* grad_out, float* g_xq, float* g_xk, float* g_xv, float* g_Wp, float* g_bp,
                                          float* g_gamma_p, float* g_beta_p, float* g_W3C, float* g_b3C, float* g_gamma_c, float* g_beta_c, float* g_Wa, float* g_ba,
                                          float* g_gamma_g, float* g_beta_g, float* g_Wb, float* g_bb, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!pw_shape_ok(n, K, C)) return CBL_ERR_UNSUPPORTED;
    if (!consts || !bnc_stats || !workspace || !grad_out) return CBL_ERR_BAD_ARG;
    if (workspace_bytes < cbl_pt_layer_wide_workspace_bytes(n, K, C)) return CBL_ERR_WORKSPACE;
    const PwWs ws = pw_workspace(static_cast<char*>(workspace), n, K, C);
    hipStream_t st = cbl_stream(stream);
    const long long np = (long long)n * K;
    const int G = C / 8, WN = 3 * G + G * G;
    const unsigned gp = pt_pair_grid(np);
    int rc;
    // the two scatters (d x_k, d x_v) of the wide kernels are float atomics: zero their targets
    if (g_xv == g_xk + (size_t)n * C) {                             // one buffer (the Python mirror allocates them together): one fill
        if (hipMemsetAsync(g_xk, 0, 2 * sizeof(float) * (size_t)n * C, st) != hipSuccess) return cbl_status();
    } else if (hipMemsetAsync(g_xk, 0, sizeof(float) * (size_t)n * C, st) != hipSuccess || hipMemsetAsync(g_xv, 0, sizeof(float) * (size_t)n * C, st) != hipSuccess) return cbl_status();
    // aggregation backward with the softmax backward inside: d x_v, its share of d p1 / d W3C / d b3C, d logits
    if ((rc = cbl_attn_agg_softmax_backward(n, K, C, G, x_v, idx, p1, W3C, b3C, a, grad_out, g_xv, ws.gp1a, ws.w3c2, ws.b3c2, ws.glogit, ws.attn, ws.attn_bytes, stream))) return rc;
    // Linear(G, G), ReLU, BN_g backward
    const long long nbatch = (np + 256 / G - 1) / (256 / G);        // a workgroup takes whole batches of 256 / G pairs
    const unsigned gn = (unsigned)(nbatch < PW_NROWS ? (nbatch > 0 ? nbatch : 1) : PW_NROWS);
#define PW_NBWD(GG) hipLaunchKernelGGL(pw_narrow_bwd_kernel<GG>, dim3(gn), dim3(256), sizeof(float) * (G * G + 4 * 256), st, np, w2, consts, Wb, ws.glogit, ws.pre, ws.part_n)
    if (G == 16) { PW_NBWD(16); } else if (G == 32) { PW_NBWD(32); } else { PW_NBWD(64); }
    hipLaunchKernelGGL(pt_bn_bwd_finalize_kernel, dim3(cbl_div_up(G, 16)), dim3(PT_FIN_THREADS), 0, st, (int)gn, WN, 0, G, G, np, ws.part_n, gamma_g, consts + PW_CST_G, 64,
                       ws.bc_g, 64, g_gamma_g, g_beta_g, consts + PW_FS_G, ws.gba2);
    hipLaunchKernelGGL(pw_bn_apply_kernel, dim3(cbl_grid_for(np * G, 256, 1024)), dim3(256), 0, st, np * G, G, ws.pre, w2, ws.bc_g, ws.gw2);
    // the C-wide backward: d x_q, d x_k, its share of d p1 / d W3C / d b3C, BN_c's and Wa's gradients
    if ((rc = cbl_attn_w2_backward(n, K, C, G, x_q, x_k, idx, p1, W3C, b3C, gamma_c, beta_c, bnc_stats, bnc_stats + C, Wa, ws.gw2, g_xq, g_xk, ws.gp1b,
                                   ws.w3c2 + 3 * (size_t)C, ws.b3c2 + C, g_gamma_c, g_beta_c, g_Wa, g_ba, ws.attn, ws.attn_bytes, stream))) return rc;
    // p chain backward over the sum of the two d p1
    hipLaunchKernelGGL(pt_pchain_bwd_kernel, dim3(gp), dim3(PT_NARROW_BLOCK), 0, st, np, p_r, p0, p1, ws.gp1a, consts, ws.part_d, (const float*)ws.gp1b);
    hipLaunchKernelGGL(pt_pchain_epilogue_kernel, dim3(1), dim3(PT_FIN_THREADS), 0, st, (int)gp, ws.part_d, np, consts, gamma_p, g_Wp, g_bp, g_gamma_p, g_beta_p);
    PtSumSegs segs;
    segs.n = 4;
    segs.s[0] = PtSumSeg{ws.part_n, g_Wb, (int)gn, WN, 2 * G, G * G};
    segs.s[1] = PtSumSeg{ws.part_n, g_bb, (int)gn, WN, 2 * G + G * G, G};
    segs.s[2] = PtSumSeg{ws.w3c2, g_W3C, 2, 3 * C, 0, 3 * C};       // the two uses of pe: aggregation and w
    segs.s[3] = PtSumSeg{ws.b3c2, g_b3C, 2, C, 0, C};
    unsigned nblk = 0;
    for (int q = 0; q < segs.n; q++) nblk += (unsigned)((segs.s[q].count + 15) / 16);
    hipLaunchKernelGGL(pt_sum_rows_kernel, dim3(nblk), dim3(PT_FIN_THREADS), 0, st, segs);
    return cbl_status();
}
#endif  // CBL_HOST_WAVE_EMULATION
