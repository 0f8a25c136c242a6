// a4, dense part: the Linear layers INSIDE the vector attention act on (n*K) rows with tiny feature widths —
//   linear_p: Linear(3,3), Linear(3,C)      linear_w: Linear(C, C/8), Linear(C/8, C/8)      /root/reference/pytorch/model/blocks.py:23-28,38-40
// i.e. GEMMs of shape (163 840 x 3) @ (3 x 3) or a weight gradient (8 x 163 840) @ (163 840 x 64).  A tiled GEMM library has no good
// kernel for that (rocBLAS: 0.4-0.5 ms per call at n*K = 163 840, profiles/): they are streaming problems — read the rows once, do a
// few FMAs per element — so they are written as such:
//   forward / input gradient:  one lane per output element, weights in LDS (row stride padded), the row's inputs read through L1
//                              (the C_out lanes of a row read the same addresses);
//   weight / bias gradient:    a workgroup stages tiles of rows of x and dy in LDS, every lane owns a few (c_out, c_in) pairs and
//                              accumulates over the workgroup's rows in registers; the <= 768 workgroup partials are summed in fp64
//                              by a second small kernel (plain stores: no pre-zeroed outputs, no atomics).
// Widths outside 16 / 32 / 48 / 64: fp32 FMA chains in index order.  Widths 16 / 32 / 48 / 64 (row_linear_mfma_kernel, row_linear_wgrad_mfma_kernel, the
// triple_linear kernels): v_mfma_f32_16x16x4_f32 chains — the contraction index is walked four at a time in the MFMA's own order (a row's quarters per
// lane), the weight gradient sums the rows four per step and the waves' partial tiles through LDS in wave order.  Either way the results differ from a GEMM
// library's by summation order only (tests: 1e-4 of the largest element).  row_linear_wgrad_mfma_kernel<64,64> keeps 2 x (4096 + 64) floats = 33 KB of LDS
// (the four waves' tiles are combined two at a time).
#include "cbl_common.h"
#include <stdlib.h>

namespace {

constexpr int SL_BLOCK = 256;
constexpr int SL_MAX_W = 4096;                 // c_in * c_out handled here (weights + padding must fit LDS comfortably)

// y[r, c] = sum_a x[r, a] * W(c, a) (+ b[c]);   W(c, a) = w[c * cin + a]  (TRANS = false: y = x W^T, nn.Linear)
//                                                        w[a * cout + c]  (TRANS = true:  y = x W,   the input gradient dy -> dx)
template <bool TRANS, bool VEC>                  // VEC: c_in % 4 == 0 and 16-byte aligned rows -> float4 reads of the row
__global__ __launch_bounds__(SL_BLOCK) void skinny_linear_kernel(long long rows, int cin, int cout, const float* __restrict__ x,
                                                                 const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ y)
{
    extern __shared__ float ws[];                                   // [cout][cin + 1]
    const int ld = cin + 1;
    for (int e = threadIdx.x; e < cin * cout; e += SL_BLOCK) {
        const int c = TRANS ? e % cout : e / cin, a = TRANS ? e / cout : e % cin;        // coalesced read of w
        ws[c * ld + a] = w[e];
    }
    __syncthreads();
    const long long total = rows * cout;
    for (long long e = (long long)blockIdx.x * SL_BLOCK + threadIdx.x; e < total; e += (long long)gridDim.x * SL_BLOCK) {
        const long long r = e / cout; const int c = (int)(e - r * cout);
        const float* __restrict__ xr = x + r * cin;
        const float* wr = ws + c * ld;
        float acc = b ? b[c] : 0.f;
        if (VEC) {
            for (int a = 0; a < cin; a += 4) {
                const float4 v = *reinterpret_cast<const float4*>(xr + a);
                acc += v.x * wr[a]; acc += v.y * wr[a + 1]; acc += v.z * wr[a + 2]; acc += v.w * wr[a + 3];
            }
        } else {
            for (int a = 0; a < cin; a++) acc += xr[a] * wr[a];
        }
        y[e] = acc;
    }
}

// partial[b][c*cin + a] = sum over workgroup b's rows of dy[r, c] * x[r, a];   partial[b][cin*cout + c] = sum of dy[r, c]
constexpr int SL_TILE = 64;                                         // rows per LDS tile
constexpr int SL_WGRAD_BLOCKS = 768;
__global__ __launch_bounds__(SL_BLOCK) void skinny_linear_wgrad_kernel(long long rows, int cin, int cout, const float* __restrict__ x,
                                                                       const float* __restrict__ dy, float* __restrict__ partial, int want_bias)
{
    extern __shared__ float tile[];                                 // x tile [SL_TILE][cin + 1], dy tile [SL_TILE][cout + 1]
    const int lx = cin + 1, ly = cout + 1;
    float* xs = tile; float* ys = tile + SL_TILE * lx;
    constexpr int PAIRS = SL_MAX_W / SL_BLOCK;                      // (c, a) pairs per lane
    const int npairs = cin * cout;
    float acc[PAIRS], accb = 0.f;
#pragma unroll
    for (int j = 0; j < PAIRS; j++) acc[j] = 0.f;
    const long long ntiles = (rows + SL_TILE - 1) / SL_TILE;
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long long r0 = t * SL_TILE;
        const int nr = (int)min((long long)SL_TILE, rows - r0);
        __syncthreads();                                            // previous tile fully consumed
        for (int e = threadIdx.x; e < nr * cin; e += SL_BLOCK) xs[(e / cin) * lx + e % cin] = x[r0 * cin + e];
        for (int e = threadIdx.x; e < nr * cout; e += SL_BLOCK) ys[(e / cout) * ly + e % cout] = dy[r0 * cout + e];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PAIRS; j++) {
            const int pidx = threadIdx.x + j * SL_BLOCK;
            if (pidx < npairs) {
                const int c = pidx / cin, a = pidx - c * cin;
                float s = 0.f;
                for (int r = 0; r < nr; r++) s += ys[r * ly + c] * xs[r * lx + a];
                acc[j] += s;
            }
        }
        if (want_bias && threadIdx.x < cout) {
            float s = 0.f;
            for (int r = 0; r < nr; r++) s += ys[r * ly + threadIdx.x];
            accb += s;
        }
    }
    float* mine = partial + (size_t)blockIdx.x * (npairs + cout);
#pragma unroll
    for (int j = 0; j < PAIRS; j++) {
        const int pidx = threadIdx.x + j * SL_BLOCK;
        if (pidx < npairs) mine[pidx] = acc[j];
    }
    if (threadIdx.x < cout) mine[npairs + threadIdx.x] = accb;
}

// dw / db = sum of the workgroup partials (fp64), 16 outputs x 16 slices per workgroup; plain stores: no pre-zeroed outputs, no atomics
__global__ __launch_bounds__(256) void skinny_linear_wgrad_finalize_kernel(int npairs, int cout, int nblocks, const float* __restrict__ partial,
                                                                           float* __restrict__ dw, float* __restrict__ db)
{
    __shared__ double red[16][16];
    const int e = blockIdx.x * 16 + (threadIdx.x & 15), js = threadIdx.x >> 4, total = npairs + cout;
    double a = 0.0;
    if (e < total) {
#pragma unroll 8
        for (int b = js; b < nblocks; b += 16) a += (double)partial[(size_t)b * total + e];
    }
    red[js][threadIdx.x & 15] = a;
    __syncthreads();
    if (js == 0 && e < total) {
        double s0 = 0.0;
        for (int j = 0; j < 16; j++) s0 += red[j][threadIdx.x & 15];
        if (e < npairs) dw[e] = (float)s0;
        else if (db) db[e - npairs] = (float)s0;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// Widths 16 / 32 / 48 / 64 on the matrix cores (v_mfma_f32_16x16x4_f32: exact f32, the f32 vector rate with 1/16 of the instructions).
// The q / k / v Linear(C, C) layers of the two full-resolution stages (C = 32, 64) run over n = 10^4 .. 10^5 rows: the streaming kernel above
// spends 26 us on 40960 x 64 x 64 (0.8 TB/s of its 21 MB), its weight gradient 41 us.
//   out (rows, ND) = in (rows, KD) . B,  B[k][n] = WT ? W[k * ND + n] : W[n * KD + k]     (forward: W (cout, cin), WT = false; input gradient: WT = true)
// One wave owns 16 rows per trip: lane (row = l % 16, kq = l / 16) loads the CONTIGUOUS quarter kq of its row (KD/4 floats, 16-byte loads — the
// contraction index may be walked in any order, so step s of lane-quarter kq stands for k = kq KD/4 + s) and feeds it as the A operand; the B
// operands of all ND/16 column tiles sit in registers for the whole launch; the next trip's rows are requested before this trip's MFMAs.
using rl_f32x4 = __attribute__((ext_vector_type(4))) float;

template <int KD, int ND, bool WT>
__global__ __launch_bounds__(256) void row_linear_mfma_kernel(long long rows, const float* __restrict__ in, const float* __restrict__ W, const float* __restrict__ bias,
                                                              float* __restrict__ out)
{
    constexpr int KC = KD / 4, NT = ND / 16;
    const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
    float bw[NT][KC];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int s2 = 0; s2 < KC; s2++) {
            const int k = KC * kq + s2, n = 16 * t + row;
            bw[t][s2] = WT ? W[(size_t)k * ND + n] : W[(size_t)n * KD + k];
        }
    float bv[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) bv[t] = bias ? bias[16 * t + row] : 0.f;
    const long long ntiles = (rows + 15) / 16;
    const long long stride = (long long)gridDim.x * 4;
    long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    float4 a[KC / 4];
    auto load = [&](long long tl) {
        const long long r = min(tl * 16 + row, rows - 1);
        const float4* src = reinterpret_cast<const float4*>(in + r * KD + KC * kq);
#pragma unroll
        for (int v = 0; v < KC / 4; v++) a[v] = src[v];
    };
    if (tile < ntiles) load(tile);
    for (; tile < ntiles; tile += stride) {
        float av[KC];
#pragma unroll
        for (int v = 0; v < KC / 4; v++) { av[4 * v] = a[v].x; av[4 * v + 1] = a[v].y; av[4 * v + 2] = a[v].z; av[4 * v + 3] = a[v].w; }
        if (tile + stride < ntiles) load(tile + stride);
        rl_f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = rl_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < KC; s2++)
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s2], bw[t][s2], acc[t], 0, 0, 0);
        // D[4 (lane / 16) + r][lane % 16] in acc[t][r]
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const long long orow = tile * 16 + 4 * kq + r;
            if (orow < rows) {
#pragma unroll
                for (int t = 0; t < NT; t++) out[orow * ND + 16 * t + row] = acc[t][r] + bv[t];
            }
        }
    }
}

// grad_weight (COUT, CIN) = grad_y^T . x and grad_bias = column sums of grad_y: the contraction runs over the ROWS, four per MFMA step
// (A[m][k] = grad_y[r0 + k][16 tm + m], B[k][n] = x[r0 + k][16 tn + n]: 64-byte row segments per 16 lanes).  A wave walks its share of the rows with
// all (COUT/16) x (CIN/16) tiles of the result in registers; waves -> LDS -> one partial per workgroup, summed by the finalize kernel above.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void row_linear_wgrad_mfma_kernel(long long rows, const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ partial,
                                                                    int want_bias)
{
    constexpr int MT = COUT / 16, NTI = CIN / 16;
    __shared__ float red[2][COUT * CIN + COUT];                     // two waves' tiles at a time (waves 2, 3 first, then 0 + 2 and 1 + 3): 33 KB at 64 x 64, not 66
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, kq = lane >> 4;
    rl_f32x4 acc[MT][NTI];
#pragma unroll
    for (int tm = 0; tm < MT; tm++)
#pragma unroll
        for (int tn = 0; tn < NTI; tn++) acc[tm][tn] = rl_f32x4{0.f, 0.f, 0.f, 0.f};
    float sb[MT];
#pragma unroll
    for (int tm = 0; tm < MT; tm++) sb[tm] = 0.f;
    const long long nsteps = (rows + 3) / 4;
    const long long gw = (long long)gridDim.x * 4;
    for (long long st = (long long)blockIdx.x * 4 + wave; st < nsteps; st += gw) {
        const long long r = st * 4 + kq;
        const bool ok = r < rows;
        const long long rc = ok ? r : rows - 1;
        float a[MT], b[NTI];
#pragma unroll
        for (int tm = 0; tm < MT; tm++) { a[tm] = gy[rc * COUT + 16 * tm + col]; a[tm] = ok ? a[tm] : 0.f; sb[tm] += a[tm]; }
#pragma unroll
        for (int tn = 0; tn < NTI; tn++) b[tn] = x[rc * CIN + 16 * tn + col];
#pragma unroll
        for (int tm = 0; tm < MT; tm++)
#pragma unroll
            for (int tn = 0; tn < NTI; tn++) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
    }
    // D[m = 4 (lane / 16) + r][n = lane % 16] of tile (tm, tn) = grad_weight[16 tm + m][16 tn + n]
    float bsum[MT];
#pragma unroll
    for (int tm = 0; tm < MT; tm++) { float v = sb[tm]; v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); bsum[tm] = v; }
    float* slot = red[wave & 1];
    if (wave >= 2) {
#pragma unroll
        for (int tm = 0; tm < MT; tm++)
#pragma unroll
            for (int tn = 0; tn < NTI; tn++)
#pragma unroll
                for (int r = 0; r < 4; r++) slot[(16 * tm + 4 * kq + r) * CIN + 16 * tn + col] = acc[tm][tn][r];
#pragma unroll
        for (int tm = 0; tm < MT; tm++) if (kq == 0) slot[COUT * CIN + 16 * tm + col] = bsum[tm];
    }
    __syncthreads();
    if (wave < 2) {                                                  // wave w adds wave w + 2's tile to its own, in place (each entry has one owner lane)
#pragma unroll
        for (int tm = 0; tm < MT; tm++)
#pragma unroll
            for (int tn = 0; tn < NTI; tn++)
#pragma unroll
                for (int r = 0; r < 4; r++) { float& e = slot[(16 * tm + 4 * kq + r) * CIN + 16 * tn + col]; e = acc[tm][tn][r] + e; }
#pragma unroll
        for (int tm = 0; tm < MT; tm++) if (kq == 0) { float& e = slot[COUT * CIN + 16 * tm + col]; e = bsum[tm] + e; }
    }
    __syncthreads();
    float* mine = partial + (size_t)blockIdx.x * (COUT * CIN + COUT);
    for (int e = threadIdx.x; e < COUT * CIN + COUT; e += 256) {
        const float sum = red[0][e] + red[1][e];
        if (e < COUT * CIN || want_bias) mine[e] = sum;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// Widths between the multiples of 16: the TransitionDown layers' Linear(3 + C, C') over the grouped (m * nsample) rows (pytorch/model/blocks.py:62-76; 35 -> 64 at
// the first down-sampling: 163 840 rows per 40960-point scene).  The streaming kernel above spends one lane per OUTPUT element with c_in scalar loads and LDS
// reads each: 100 us forward, 60 us input gradient, 80 us weight gradient per scene (`skinny_linear_kernel<false, false>` and friends in the network's profile).
// The same MFMA walks as above over operands padded with zeros IN REGISTERS: the ragged side (k < kvalid of the input rows, n < nvalid of the output rows)
// is read and written with the true row strides, 4 bytes at a time (rows of 35 floats are not 16-byte aligned).
template <int KD, int ND, bool WT>
__global__ __launch_bounds__(256) void row_linear_ragged_mfma_kernel(long long rows, int kvalid, int nvalid, int w_ld, const float* __restrict__ in, const float* __restrict__ W,
                                                                     const float* __restrict__ bias, float* __restrict__ out)
{
    constexpr int KC = KD / 4, NT = ND / 16;
    const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
    float bw[NT][KC];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int s2 = 0; s2 < KC; s2++) {
            const int k = KC * kq + s2, n = 16 * t + row;
            const bool ok = k < kvalid && n < nvalid;
            const int kc = min(k, kvalid - 1), nc = min(n, nvalid - 1);
            const float v = WT ? W[(size_t)kc * w_ld + nc] : W[(size_t)nc * w_ld + kc];
            bw[t][s2] = ok ? v : 0.f;
        }
    float bv[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) bv[t] = (bias && 16 * t + row < nvalid) ? bias[16 * t + row] : 0.f;
    const long long ntiles = (rows + 15) / 16;
    const long long stride = (long long)gridDim.x * 4;
    long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    float a[KC];
    const bool vec = kvalid == KD && ((reinterpret_cast<uintptr_t>(in) & 15u) == 0);   // the input side is whole (the input gradient: grad_y rows): 16-byte loads
    auto load = [&](long long tl) {
        const long long r = min(tl * 16 + row, rows - 1);
        const float* src = in + r * kvalid;
        if (vec) {
            const float4* s4 = reinterpret_cast<const float4*>(src + KC * kq);
#pragma unroll
            for (int v = 0; v < KC / 4; v++) { const float4 t = s4[v]; a[4 * v] = t.x; a[4 * v + 1] = t.y; a[4 * v + 2] = t.z; a[4 * v + 3] = t.w; }
            return;
        }
#pragma unroll
        for (int j = 0; j < KC; j++) {
            const int k = KC * kq + j;
            const float v = src[min(k, kvalid - 1)];                 // clamped, unconditional: all KC loads in flight together
            a[j] = k < kvalid ? v : 0.f;
        }
    };
    if (tile < ntiles) load(tile);
    for (; tile < ntiles; tile += stride) {
        float av[KC];
#pragma unroll
        for (int j = 0; j < KC; j++) av[j] = a[j];
        if (tile + stride < ntiles) load(tile + stride);
        rl_f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = rl_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < KC; s2++)
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s2], bw[t][s2], acc[t], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const long long orow = tile * 16 + 4 * kq + r;
            if (orow < rows) {
#pragma unroll
                for (int t = 0; t < NT; t++) if (16 * t + row < nvalid) out[orow * nvalid + 16 * t + row] = acc[t][r] + bv[t];
            }
        }
    }
}

// weight / bias gradient with a ragged c_in: x rows of `cin` floats (CINP = cin rounded up to 16), partial rows of the true cin * COUT + COUT floats
template <int CINP, int COUT>
__global__ __launch_bounds__(256) void row_linear_wgrad_ragged_mfma_kernel(long long rows, int cin, const float* __restrict__ x, const float* __restrict__ gy,
                                                                           float* __restrict__ partial, int want_bias)
{
    constexpr int MT = COUT / 16, NTI = CINP / 16;
    __shared__ float red[2][COUT * CINP + COUT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, kq = lane >> 4;
    rl_f32x4 acc[MT][NTI];
#pragma unroll
    for (int tm = 0; tm < MT; tm++)
#pragma unroll
        for (int tn = 0; tn < NTI; tn++) acc[tm][tn] = rl_f32x4{0.f, 0.f, 0.f, 0.f};
    float sb[MT];
#pragma unroll
    for (int tm = 0; tm < MT; tm++) sb[tm] = 0.f;
    const long long nsteps = (rows + 3) / 4;
    const long long gw = (long long)gridDim.x * 4;
    for (long long st = (long long)blockIdx.x * 4 + wave; st < nsteps; st += gw) {
        const long long r = st * 4 + kq;
        const bool ok = r < rows;
        const long long rc = ok ? r : rows - 1;
        float a[MT], b[NTI];
#pragma unroll
        for (int tm = 0; tm < MT; tm++) { a[tm] = gy[rc * COUT + 16 * tm + col]; a[tm] = ok ? a[tm] : 0.f; sb[tm] += a[tm]; }
#pragma unroll
        for (int tn = 0; tn < NTI; tn++) { const int n = 16 * tn + col; const float v = x[rc * cin + min(n, cin - 1)]; b[tn] = n < cin ? v : 0.f; }
#pragma unroll
        for (int tm = 0; tm < MT; tm++)
#pragma unroll
            for (int tn = 0; tn < NTI; tn++) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
    }
    float bsum[MT];
#pragma unroll
    for (int tm = 0; tm < MT; tm++) { float v = sb[tm]; v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); bsum[tm] = v; }
    float* slot = red[wave & 1];
    if (wave >= 2) {
#pragma unroll
        for (int tm = 0; tm < MT; tm++)
#pragma unroll
            for (int tn = 0; tn < NTI; tn++)
#pragma unroll
                for (int r = 0; r < 4; r++) slot[(16 * tm + 4 * kq + r) * CINP + 16 * tn + col] = acc[tm][tn][r];
#pragma unroll
        for (int tm = 0; tm < MT; tm++) if (kq == 0) slot[COUT * CINP + 16 * tm + col] = bsum[tm];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int tm = 0; tm < MT; tm++)
#pragma unroll
            for (int tn = 0; tn < NTI; tn++)
#pragma unroll
                for (int r = 0; r < 4; r++) { float& e = slot[(16 * tm + 4 * kq + r) * CINP + 16 * tn + col]; e = acc[tm][tn][r] + e; }
#pragma unroll
        for (int tm = 0; tm < MT; tm++) if (kq == 0) { float& e = slot[COUT * CINP + 16 * tm + col]; e = bsum[tm] + e; }
    }
    __syncthreads();
    float* mine = partial + (size_t)blockIdx.x * ((size_t)COUT * cin + COUT);
    for (int e = threadIdx.x; e < COUT * CINP + COUT; e += 256) {
        const float sum = red[0][e] + red[1][e];
        if (e < COUT * CINP) { const int m = e / CINP, n = e - m * CINP; if (n < cin) mine[(size_t)m * cin + n] = sum; }
        else if (want_bias) mine[(size_t)COUT * cin + (e - COUT * CINP)] = sum;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// The three projections of PointTransformerLayer (blocks.py:33: x_q, x_k, x_v = linear_q(x), linear_k(x), linear_v(x); C = 32 / 64) as ONE launch
// per direction: forward with blockIdx.y = the projection; input gradient d x = d x_q Wq + d x_k Wk + d x_v Wv accumulated in the MFMA accumulators
// of one pass over the three gradient tensors (weights in LDS: 3 C^2 operands do not fit the registers), weight / bias gradients with
// blockIdx.y = the projection.  (Three Linear layers cost 3 + 3 + 6 launches and two adds: ~130 us of a 500 us layer step at (40960, 64).)
struct RlTriple { const float* in[3]; const float* w[3]; const float* b[3]; float* out[3]; };

// One wave per SIMD (a workgroup per CU, the whole register file): the 3 C^2 / 64 B operands of a lane stay in registers for the launch, a
// 16-row tile of x is loaded once for the three projections, the next tile's rows are requested before this tile's MFMAs.  (With blockIdx.y =
// the projection every wave loaded 64 operands for ~5 tiles of work: 24 us at (40960, 64).)
template <int C>
__global__ __launch_bounds__(256, 1) void triple_linear_forward_kernel(long long rows, const float* __restrict__ in, RlTriple t3)
{
    constexpr int KC = C / 4, NT = C / 16;
    const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
    float bw[3][KC][NT], bv[3][NT];                                  // B[k][n] = W_p[n][k], k = KC kq + s
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int t = 0; t < NT; t++) {
#pragma unroll
            for (int s2 = 0; s2 < KC; s2++) bw[p][s2][t] = t3.w[p][(size_t)(16 * t + row) * C + KC * kq + s2];
            bv[p][t] = t3.b[p] ? t3.b[p][16 * t + row] : 0.f;
        }
    const long long ntiles = (rows + 15) / 16, stride = (long long)gridDim.x * 4;
    long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    float4 a[KC / 4];
    auto load = [&](long long tl) {
        const long long r = min(tl * 16 + row, rows - 1);
        const float4* src = reinterpret_cast<const float4*>(in + r * C + KC * kq);
#pragma unroll
        for (int v = 0; v < KC / 4; v++) a[v] = src[v];
    };
    if (tile < ntiles) load(tile);
    for (; tile < ntiles; tile += stride) {
        float av[KC];
#pragma unroll
        for (int v = 0; v < KC / 4; v++) { av[4 * v] = a[v].x; av[4 * v + 1] = a[v].y; av[4 * v + 2] = a[v].z; av[4 * v + 3] = a[v].w; }
        if (tile + stride < ntiles) load(tile + stride);
#pragma unroll
        for (int p = 0; p < 3; p++) {
            rl_f32x4 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = rl_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < KC; s2++)
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s2], bw[p][s2][t], acc[t], 0, 0, 0);
            float* __restrict__ out = t3.out[p];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const long long orow = tile * 16 + 4 * kq + r;
                if (orow < rows) {
#pragma unroll
                    for (int t = 0; t < NT; t++) out[orow * C + 16 * t + row] = acc[t][r] + bv[p][t];
                }
            }
        }
    }
}

// d x (rows, C) = sum over the three projections of d y_p (rows, C) . W_p (C, C): one accumulator set per 16-row tile.  One wave per SIMD
// (a workgroup per CU, the whole register file): the 3 C^2 / 64 B operands of a lane stay in registers for the launch, the next tile's
// rows are requested before this tile's 3 C / 4 x C / 16 MFMAs.  (A first version with the weights in LDS and a tile per wave spent its time in
// its prologue: 49 us for 0.5 M MFMAs.)
template <int C>
__global__ __launch_bounds__(256, 1) void triple_linear_dgrad_kernel(long long rows, RlTriple t3, float* __restrict__ gx)
{
    constexpr int KC = C / 4, NT = C / 16;
    const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
    float bw[3][KC][NT];                                             // B[k][n] = W_p[k][n], k = KC kq + s (the contraction runs over the projection's outputs)
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int s2 = 0; s2 < KC; s2++)
#pragma unroll
            for (int t = 0; t < NT; t++) bw[p][s2][t] = t3.w[p][(size_t)(KC * kq + s2) * C + 16 * t + row];
    const long long ntiles = (rows + 15) / 16, stride = (long long)gridDim.x * 4;
    long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    float4 a[3][KC / 4];
    auto load = [&](long long tl) {
        const long long r = min(tl * 16 + row, rows - 1);
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const float4* src = reinterpret_cast<const float4*>(t3.in[p] + r * C + KC * kq);
#pragma unroll
            for (int v = 0; v < KC / 4; v++) a[p][v] = src[v];
        }
    };
    if (tile < ntiles) load(tile);
    for (; tile < ntiles; tile += stride) {
        float av[3][KC];
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int v = 0; v < KC / 4; v++) { av[p][4 * v] = a[p][v].x; av[p][4 * v + 1] = a[p][v].y; av[p][4 * v + 2] = a[p][v].z; av[p][4 * v + 3] = a[p][v].w; }
        if (tile + stride < ntiles) load(tile + stride);
        rl_f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = rl_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int s2 = 0; s2 < KC; s2++)
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[p][s2], bw[p][s2][t], acc[t], 0, 0, 0);
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const long long orow = tile * 16 + 4 * kq + rr;
            if (orow < rows) {
#pragma unroll
                for (int t = 0; t < NT; t++) gx[orow * C + 16 * t + row] = acc[t][rr];
            }
        }
    }
}

// weight / bias gradients of the three projections: row_linear_wgrad_mfma_kernel's pass with blockIdx.y = the projection (partials side by side)
template <int C>
__global__ __launch_bounds__(256) void triple_linear_wgrad_kernel(long long rows, const float* __restrict__ x, RlTriple t3, float* __restrict__ partial)
{
    constexpr int MT = C / 16, NTI = C / 16, WIDTH = C * C + C;
    __shared__ float red[2][WIDTH];                                 // two slots for four waves (33 KB at C = 64, was 66.5): waves 2, 3 store, waves 0, 1 add theirs on top
    const float* __restrict__ gy = t3.in[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, kq = lane >> 4;
    rl_f32x4 acc[MT][NTI];
#pragma unroll
    for (int tm = 0; tm < MT; tm++)
#pragma unroll
        for (int tn = 0; tn < NTI; tn++) acc[tm][tn] = rl_f32x4{0.f, 0.f, 0.f, 0.f};
    float sb[MT];
#pragma unroll
    for (int tm = 0; tm < MT; tm++) sb[tm] = 0.f;
    const long long nsteps = (rows + 3) / 4;
    const long long gw = (long long)gridDim.x * 4;
    for (long long st = (long long)blockIdx.x * 4 + wave; st < nsteps; st += gw) {
        const long long r = st * 4 + kq;
        const bool ok = r < rows;
        const long long rc = ok ? r : rows - 1;
        float a[MT], b[NTI];
#pragma unroll
        for (int tm = 0; tm < MT; tm++) { a[tm] = gy[rc * C + 16 * tm + col]; a[tm] = ok ? a[tm] : 0.f; sb[tm] += a[tm]; }
#pragma unroll
        for (int tn = 0; tn < NTI; tn++) b[tn] = x[rc * C + 16 * tn + col];
#pragma unroll
        for (int tm = 0; tm < MT; tm++)
#pragma unroll
            for (int tn = 0; tn < NTI; tn++) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
    }
#pragma unroll
    for (int tm = 0; tm < MT; tm++) { sb[tm] += __shfl_xor(sb[tm], 16); sb[tm] += __shfl_xor(sb[tm], 32); }
    float* slot = red[wave & 1];                                     // every element of a slot belongs to one lane of the wave that owns it
    if (wave >= 2) {
#pragma unroll
        for (int tm = 0; tm < MT; tm++) {
#pragma unroll
            for (int tn = 0; tn < NTI; tn++)
#pragma unroll
                for (int r = 0; r < 4; r++) slot[(16 * tm + 4 * kq + r) * C + 16 * tn + col] = acc[tm][tn][r];
            if (kq == 0) slot[C * C + 16 * tm + col] = sb[tm];
        }
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int tm = 0; tm < MT; tm++) {
#pragma unroll
            for (int tn = 0; tn < NTI; tn++)
#pragma unroll
                for (int r = 0; r < 4; r++) slot[(16 * tm + 4 * kq + r) * C + 16 * tn + col] += acc[tm][tn][r];
            if (kq == 0) slot[C * C + 16 * tm + col] += sb[tm];
        }
    }
    __syncthreads();
    float* mine = partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * WIDTH;
    for (int e = threadIdx.x; e < WIDTH; e += 256) mine[e] = red[0][e] + red[1][e];
}

__global__ __launch_bounds__(1024) void triple_linear_wgrad_finalize_kernel(int C, int nblocks, const float* __restrict__ partial, RlTriple t3)
{
    __shared__ double red[64][16];
    const int width = C * C + C, e = blockIdx.x * 16 + (threadIdx.x & 15), js = threadIdx.x >> 4;
    const float* src = partial + (size_t)blockIdx.y * nblocks * width;
    double a = 0.0;
    if (e < width) for (int b = js; b < nblocks; b += 64) a += (double)src[(size_t)b * width + e];
    red[js][threadIdx.x & 15] = a;
    __syncthreads();
    if (js == 0 && e < width) {
        double s0 = 0.0;
        for (int j = 0; j < 64; j++) s0 += red[j][threadIdx.x & 15];
        if (e < C * C) t3.out[blockIdx.y][e] = (float)s0;
        else if (t3.b[blockIdx.y]) const_cast<float*>(t3.b[blockIdx.y])[e - C * C] = (float)s0;
    }
}

constexpr int TL_WGRAD_BLOCKS = 256;

inline bool rl_mfma_ok(int cin, int cout) { return cin % 16 == 0 && cout % 16 == 0 && cin <= 64 && cout <= 64 && cin >= 16 && cout >= 16; }
// a ragged c_in between 17 and 63 beside a c_out the matrix tiles cover: the padded-in-registers kernels
inline bool rl_ragged_ok(int cin, int cout) { return cin > 16 && cin < 64 && cin % 16 != 0 && cout % 16 == 0 && cout >= 16 && cout <= 64; }
inline int rl_up16(int v) { return (v + 15) & ~15; }

// WT = false: forward (in = x, kvalid = cin, out stride nvalid = cout);  WT = true: input gradient (in = grad_y, kvalid = cout, nvalid = cin)
template <bool WT>
int rl_ragged_launch(long long rows, int kvalid, int nvalid, int w_ld, const float* in, const float* W, const float* bias, float* out, hipStream_t st)
{
    const int kd = rl_up16(kvalid), nd = rl_up16(nvalid);
    const long long tiles = (rows + 15) / 16;
    const dim3 grid((unsigned)min((tiles + 3) / 4, (long long)2048)), blk(256);
#define CBL_RLR(KD_, ND_) if (kd == KD_ && nd == ND_) { hipLaunchKernelGGL((row_linear_ragged_mfma_kernel<KD_, ND_, WT>), grid, blk, 0, st, rows, kvalid, nvalid, w_ld, in, W, bias, out); return cbl_status(); }
    CBL_RLR(16, 32) CBL_RLR(16, 48) CBL_RLR(16, 64) CBL_RLR(32, 16) CBL_RLR(32, 32) CBL_RLR(32, 48) CBL_RLR(32, 64)
    CBL_RLR(48, 16) CBL_RLR(48, 32) CBL_RLR(48, 48) CBL_RLR(48, 64) CBL_RLR(64, 16) CBL_RLR(64, 32) CBL_RLR(64, 48) CBL_RLR(64, 64)
#undef CBL_RLR
    return CBL_ERR_UNSUPPORTED;
}

int rl_wgrad_ragged_launch(long long rows, int cin, int cout, const float* x, const float* gy, float* partial, int want_bias, int nblocks, hipStream_t st)
{
    const int cp = rl_up16(cin);
#define CBL_RLWR(CI_, CO_) if (cp == CI_ && cout == CO_) { hipLaunchKernelGGL((row_linear_wgrad_ragged_mfma_kernel<CI_, CO_>), dim3(nblocks), dim3(256), 0, st, rows, cin, x, gy, partial, want_bias); return cbl_status(); }
    CBL_RLWR(32, 16) CBL_RLWR(32, 32) CBL_RLWR(32, 48) CBL_RLWR(32, 64) CBL_RLWR(48, 16) CBL_RLWR(48, 32) CBL_RLWR(48, 48) CBL_RLWR(48, 64)
    CBL_RLWR(64, 16) CBL_RLWR(64, 32) CBL_RLWR(64, 48) CBL_RLWR(64, 64)
#undef CBL_RLWR
    return CBL_ERR_UNSUPPORTED;
}

template <bool WT>
int rl_launch(long long rows, int kd, int nd, const float* in, const float* W, const float* bias, float* out, hipStream_t st)
{
    const long long tiles = (rows + 15) / 16;
    const dim3 grid((unsigned)min((tiles + 3) / 4, (long long)2048)), blk(256);
#define CBL_RL(KD_, ND_) if (kd == KD_ && nd == ND_) { hipLaunchKernelGGL((row_linear_mfma_kernel<KD_, ND_, WT>), grid, blk, 0, st, rows, in, W, bias, out); return cbl_status(); }
    CBL_RL(16, 16) CBL_RL(16, 32) CBL_RL(16, 48) CBL_RL(16, 64) CBL_RL(32, 16) CBL_RL(32, 32) CBL_RL(32, 48) CBL_RL(32, 64)
    CBL_RL(48, 16) CBL_RL(48, 32) CBL_RL(48, 48) CBL_RL(48, 64) CBL_RL(64, 16) CBL_RL(64, 32) CBL_RL(64, 48) CBL_RL(64, 64)
#undef CBL_RL
    return CBL_ERR_UNSUPPORTED;
}

int rl_wgrad_launch(long long rows, int cin, int cout, const float* x, const float* gy, float* partial, int want_bias, int nblocks, hipStream_t st)
{
#define CBL_RLW(CI_, CO_) if (cin == CI_ && cout == CO_) { hipLaunchKernelGGL((row_linear_wgrad_mfma_kernel<CI_, CO_>), dim3(nblocks), dim3(256), 0, st, rows, x, gy, partial, want_bias); return cbl_status(); }
    CBL_RLW(16, 16) CBL_RLW(16, 32) CBL_RLW(16, 48) CBL_RLW(16, 64) CBL_RLW(32, 16) CBL_RLW(32, 32) CBL_RLW(32, 48) CBL_RLW(32, 64)
    CBL_RLW(48, 16) CBL_RLW(48, 32) CBL_RLW(48, 48) CBL_RLW(48, 64) CBL_RLW(64, 16) CBL_RLW(64, 32) CBL_RLW(64, 48) CBL_RLW(64, 64)
#undef CBL_RLW
    return CBL_ERR_UNSUPPORTED;
}

int sl_check(long long rows, int cin, int cout)
{
    if (rows < 0 || cin <= 0 || cout <= 0) return CBL_ERR_BAD_ARG;
    if ((long long)cin * cout > SL_MAX_W || cin + cout > 200) return CBL_ERR_UNSUPPORTED;     // LDS tiles: 64 rows x (c_in + c_out + 2) floats
    return CBL_OK;
}

}  // namespace

CBL_EXPORT int cbl_skinny_linear_forward(long long rows, int cin, int cout, const float* x, const float* weight, const float* bias, float* y, void* stream)
{
    const int rc = sl_check(rows, cin, cout);
    if (rc) return rc;
    if (rows == 0) return CBL_OK;
    if (!x || !weight || !y) return CBL_ERR_BAD_ARG;
    if (rl_mfma_ok(cin, cout) && cbl_host_aligned16(x)) return rl_launch<false>(rows, cin, cout, x, weight, bias, y, cbl_stream(stream));
    if (rl_ragged_ok(cin, cout)) return rl_ragged_launch<false>(rows, cin, cout, cin, x, weight, bias, y, cbl_stream(stream));
    const dim3 grid(cbl_grid_for(rows * cout, SL_BLOCK, 2048));
    const size_t lds = sizeof(float) * (size_t)cout * (cin + 1);
    if (cin % 4 == 0 && cbl_host_aligned16(x))
        hipLaunchKernelGGL((skinny_linear_kernel<false, true>), grid, dim3(SL_BLOCK), lds, cbl_stream(stream), rows, cin, cout, x, weight, bias, y);
    else
        hipLaunchKernelGGL((skinny_linear_kernel<false, false>), grid, dim3(SL_BLOCK), lds, cbl_stream(stream), rows, cin, cout, x, weight, bias, y);
    return cbl_status();
}

CBL_EXPORT int cbl_skinny_linear_backward_input(long long rows, int cin, int cout, const float* grad_y, const float* weight, float* grad_x, void* stream)
{
    const int rc = sl_check(rows, cin, cout);
    if (rc) return rc;
    if (rows == 0) return CBL_OK;
    if (!grad_y || !weight || !grad_x) return CBL_ERR_BAD_ARG;
    if (rl_mfma_ok(cin, cout) && cbl_host_aligned16(grad_y)) return rl_launch<true>(rows, cout, cin, grad_y, weight, nullptr, grad_x, cbl_stream(stream));
    if (rl_ragged_ok(cin, cout)) return rl_ragged_launch<true>(rows, cout, cin, cin, grad_y, weight, nullptr, grad_x, cbl_stream(stream));
    // dx = dy @ W: the same kernel with the roles of c_in / c_out swapped and W read transposed
    const dim3 grid(cbl_grid_for(rows * cin, SL_BLOCK, 2048));
    const size_t lds = sizeof(float) * (size_t)cin * (cout + 1);
    if (cout % 4 == 0 && cbl_host_aligned16(grad_y))
        hipLaunchKernelGGL((skinny_linear_kernel<true, true>), grid, dim3(SL_BLOCK), lds, cbl_stream(stream), rows, cout, cin, grad_y, weight, nullptr, grad_x);
    else
        hipLaunchKernelGGL((skinny_linear_kernel<true, false>), grid, dim3(SL_BLOCK), lds, cbl_stream(stream), rows, cout, cin, grad_y, weight, nullptr, grad_x);
    return cbl_status();
}

CBL_EXPORT size_t cbl_skinny_linear_workspace_bytes(int cin, int cout)
{
    if (sl_check(0, cin, cout) != CBL_OK) return 0;
    return sizeof(float) * (size_t)SL_WGRAD_BLOCKS * ((size_t)cin * cout + cout) + 256;
}

CBL_EXPORT int cbl_skinny_linear_backward_weight(long long rows, int cin, int cout, const float* x, const float* grad_y, float* grad_weight, float* grad_bias,
                                                 void* workspace, size_t workspace_bytes, void* stream)
{
    const int rc = sl_check(rows, cin, cout);
    if (rc) return rc;
    if (!grad_weight || !workspace) return CBL_ERR_BAD_ARG;
    if (workspace_bytes < cbl_skinny_linear_workspace_bytes(cin, cout)) return CBL_ERR_WORKSPACE;
    hipStream_t st = cbl_stream(stream);
    if (rows == 0) {
        (void)hipMemsetAsync(grad_weight, 0, sizeof(float) * (size_t)cin * cout, st);
        if (grad_bias) (void)hipMemsetAsync(grad_bias, 0, sizeof(float) * cout, st);
        return cbl_status();
    }
    if (!x || !grad_y) return CBL_ERR_BAD_ARG;
    const long long ntiles = (rows + SL_TILE - 1) / SL_TILE;
    const int nblocks = (int)min(ntiles, (long long)SL_WGRAD_BLOCKS);
    float* partial = reinterpret_cast<float*>(workspace);
    if (rl_mfma_ok(cin, cout)) {
        const int nb = (int)min((rows + 63) / 64, (long long)SL_WGRAD_BLOCKS);
        const int rc2 = rl_wgrad_launch(rows, cin, cout, x, grad_y, partial, grad_bias ? 1 : 0, nb, st);
        if (rc2) return rc2;
        hipLaunchKernelGGL(skinny_linear_wgrad_finalize_kernel, dim3(cbl_div_up(cin * cout + cout, 16)), dim3(256), 0, st, cin * cout, cout, nb, partial,
                           grad_weight, grad_bias);
        return cbl_status();
    }
    if (rl_ragged_ok(cin, cout)) {
        const int nb = (int)min((rows + 63) / 64, (long long)SL_WGRAD_BLOCKS);
        const int rc2 = rl_wgrad_ragged_launch(rows, cin, cout, x, grad_y, partial, grad_bias ? 1 : 0, nb, st);
        if (rc2) return rc2;
        hipLaunchKernelGGL(skinny_linear_wgrad_finalize_kernel, dim3(cbl_div_up(cin * cout + cout, 16)), dim3(256), 0, st, cin * cout, cout, nb, partial,
                           grad_weight, grad_bias);
        return cbl_status();
    }
    hipLaunchKernelGGL(skinny_linear_wgrad_kernel, dim3(nblocks), dim3(SL_BLOCK), sizeof(float) * (size_t)SL_TILE * (cin + 1 + cout + 1), st,
                       rows, cin, cout, x, grad_y, partial, grad_bias ? 1 : 0);
    hipLaunchKernelGGL(skinny_linear_wgrad_finalize_kernel, dim3(cbl_div_up(cin * cout + cout, 16)), dim3(256), 0, st, cin * cout, cout, nblocks, partial,
                       grad_weight, grad_bias);
    return cbl_status();
}

// ---- the three projections of PointTransformerLayer as one launch per direction (blocks.py:33; C = 32 | 64) ----
CBL_EXPORT size_t cbl_triple_linear_workspace_bytes(int C)
{
    if (C != 32 && C != 64) return 0;
    return sizeof(float) * 3 * (size_t)TL_WGRAD_BLOCKS * ((size_t)C * C + C) + 256;
}

CBL_EXPORT int cbl_triple_linear_forward(long long rows, int C, const float* x, const float* const* weight3, const float* const* bias3, float* const* y3, void* stream)
{
    if (C != 32 && C != 64) return CBL_ERR_UNSUPPORTED;
    if (rows <= 0 || !x || !weight3 || !y3 || !cbl_host_aligned16(x)) return CBL_ERR_BAD_ARG;
    RlTriple t3;
    for (int p = 0; p < 3; p++) { t3.in[p] = x; t3.w[p] = weight3[p]; t3.b[p] = bias3 ? bias3[p] : nullptr; t3.out[p] = y3[p]; }
    const long long tiles = (rows + 15) / 16;
    const dim3 grid((unsigned)min((tiles + 3) / 4, (long long)256)), blk(256);
    if (C == 64) hipLaunchKernelGGL(triple_linear_forward_kernel<64>, grid, blk, 0, cbl_stream(stream), rows, x, t3);
    else         hipLaunchKernelGGL(triple_linear_forward_kernel<32>, grid, blk, 0, cbl_stream(stream), rows, x, t3);
    return cbl_status();
}

/* grad_x = sum_p grad_y3[p] . weight3[p];  grad_weight3[p] = grad_y3[p]^T . x, grad_bias3[p] = column sums of grad_y3[p] (entries of grad_bias3 may be NULL) */
CBL_EXPORT int cbl_triple_linear_backward(long long rows, int C, const float* x, const float* const* weight3, const float* const* grad_y3, float* grad_x,
                                          float* const* grad_weight3, float* const* grad_bias3, void* workspace, size_t workspace_bytes, void* stream)
{
    if (C != 32 && C != 64) return CBL_ERR_UNSUPPORTED;
    if (rows <= 0 || !x || !weight3 || !grad_y3 || !grad_x || !grad_weight3 || !workspace) return CBL_ERR_BAD_ARG;
    if (workspace_bytes < cbl_triple_linear_workspace_bytes(C)) return CBL_ERR_WORKSPACE;
    for (int p = 0; p < 3; p++) if (!cbl_host_aligned16(grad_y3[p])) return CBL_ERR_BAD_ARG;
    hipStream_t st = cbl_stream(stream);
    RlTriple t3;
    for (int p = 0; p < 3; p++) { t3.in[p] = grad_y3[p]; t3.w[p] = weight3[p]; t3.b[p] = grad_bias3 ? grad_bias3[p] : nullptr; t3.out[p] = grad_weight3[p]; }
    const long long tiles = (rows + 15) / 16;
    const dim3 grid((unsigned)min((tiles + 3) / 4, (long long)256)), blk(256);
    if (C == 64) hipLaunchKernelGGL(triple_linear_dgrad_kernel<64>, grid, blk, 0, st, rows, t3, grad_x);
    else         hipLaunchKernelGGL(triple_linear_dgrad_kernel<32>, grid, blk, 0, st, rows, t3, grad_x);
    float* partial = reinterpret_cast<float*>(workspace);
    const int nb = (int)min((rows + 63) / 64, (long long)TL_WGRAD_BLOCKS);
    if (C == 64) hipLaunchKernelGGL(triple_linear_wgrad_kernel<64>, dim3(nb, 3), blk, 0, st, rows, x, t3, partial);
    else         hipLaunchKernelGGL(triple_linear_wgrad_kernel<32>, dim3(nb, 3), blk, 0, st, rows, x, t3, partial);
    hipLaunchKernelGGL(triple_linear_wgrad_finalize_kernel, dim3(cbl_div_up(C * C + C, 16), 3), dim3(1024), 0, st, C, nb, partial, t3);
    return cbl_status();
}
