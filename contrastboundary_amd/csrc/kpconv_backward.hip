// a15 backward without atomics: KPConv's gradients as a gather over the transposed neighbour table, on the matrix cores.
//   PseudoGrid math   /root/reference/tensorflow/models/local_aggregation_operators.py:681-728
//     out[i,c] = sum_kp kw[kp,c] * sum_k w[i,kp,k] * f[nbr(i,k), c],   w = influence of kernel point kp on neighbour k of point i
// Both gradients go through ONE per-target matrix:  S_j[kp,c] = sum over the pairs p = (i,k) with nbr(i,k) = j of  w[p,kp] * go[i_p,c]
//   d out / d f  :  grad_f[j,c]   = sum_kp kw[kp,c] * S_j[kp,c]
//   d out / d kw :  grad_kw[kp,c] = sum_j  f[j,c]   * S_j[kp,c]
// and S_j = W_j^T (KP x pairs) . G_j (pairs x C) is a small GEMM whose contraction runs over the target's PAIRS: the forward kernel
// transposed.  One wave owns a target row j and walks its pairs (cbl_neighbor_transpose) four per step on v_mfma_f32_16x16x4_f32
// (M = 16 kernel points, N = 16 channels x 4 tiles, k = 4 pairs): the A operand is the influence weight, ONE per lane and step, computed
// in registers by the lane that owns (kernel point, pair); the B operand is the gathered output-gradient row (column j of tile t stands
// for channel 4j + t, so a lane's four B values are one 16-byte load, as in the forward kernel); S lives in 16 accumulator registers.
// Per target the epilogue contracts S with the kernel weights (grad_f, one 16-byte store per 16 lanes) and adds f_j * S into 16
// persistent registers (grad_kw), reduced at the end over the waves of a workgroup (LDS) and the workgroups (per-workgroup partial rows +
// one small reduction kernel).  No atomics anywhere, written not accumulated, deterministic.
// Round 1 walked the query points and scattered d out / d f with one float atomic per (pair, channel): 42 M atomics on the memory side of
// the fabric, 298 us at N = 40960, K = 16, C = 64.  A first gather version on the VALU (weights rotated across 16 lanes by DPP, 128
// accumulators per lane) was bound by instruction issue at two waves per SIMD: 70 us (tools/exp/valu_rate.hip: a wave-instruction costs
// 3 clocks at that occupancy, a DPP one 5.5).
#include "cbl_common.h"
#include "wave_ops.h"
#include <stdlib.h>

namespace {

constexpr int KB_NB = 256;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// one wave per target row (four per workgroup); C % 4 == 0, rows 16-byte aligned, KP <= 16.
// partial: (gridDim.x, KP, C) per-workgroup sums of grad_kw.
template <bool GF, bool GKW, bool CLOSEST>
__global__ __launch_bounds__(KB_NB) void kpconv_bwd_csr_kernel(unsigned n0, int C, int KP, CblFastDiv dvK, const float* __restrict__ q, const float* __restrict__ s,
                                                               const float* __restrict__ f, const float* __restrict__ kpts, const float* __restrict__ kw,
                                                               float extent, int influence, const float* __restrict__ go,
                                                               const int* __restrict__ order, const int* __restrict__ inv_start, const int* __restrict__ inv_src,
                                                               float* __restrict__ gf, float* __restrict__ partial)
{
    __shared__ float red[KB_NB / 64][16][64];                        // grad_kw of the four waves: [wave][kernel point][channel of the chunk]
    __shared__ float4 kw_lds[16][16];                                // kernel weights of the chunk: [kernel point][channel quad]
    __shared__ float4 pairs_lds[KB_NB / 64][64];                     // per wave: (query point id, offset of the target from it) of the 64 pairs in hand
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kp = lane & 15;             // A row / kernel point; also B / D column j
    const int kq = lane >> 4;             // pair within a step of four; D row block
    const bool kp_ok = kp < KP;
    const float kx = kp_ok ? kpts[3 * kp] : 0.f, ky = kp_ok ? kpts[3 * kp + 1] : 0.f, kz = kp_ok ? kpts[3 * kp + 2] : 0.f;
    const float inv_extent = 1.0f / extent;
    const unsigned nwg = (n0 + 3) >> 2;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const bool cok = c0 + 4 * kp < C;
        const int cb = cok ? c0 + 4 * kp : c0;                       // lanes beyond the last channel read the chunk's first four and write nothing
        // kernel weights of a lane's accumulator elements (tile t, row r <-> kernel point 4*kq + r, channel cb + t) wait in LDS for the epilogue
        // of a target (sixteen registers less: four waves per SIMD), zero where the kernel point or the channel does not exist
        __syncthreads();
        {
            const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
            const bool on = row < KP && c0 + 4 * col < C;
            kw_lds[row][col] = on ? *reinterpret_cast<const float4*>(kw + (size_t)row * C + c0 + 4 * col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        f32x4 gk[4];
#pragma unroll
        for (int t = 0; t < 4; t++) gk[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        // A target is five dependent round trips (sequence slot -> row id and list bounds -> pairs and coordinates -> query coordinates ->
        // gradient rows: ~1100 clocks each, the tables were written by other XCDs) in front of ~500 clocks of matrix work; PMC: 56 % of the
        // wave cycles were s_waitcnt.  The first three are prefetched: the bounds two targets ahead, the first 64 pairs and the target's
        // coordinates one target ahead (values only, nothing branches on them before their own trip).
        const unsigned vend = 8 * cbl_xcd_per(nwg), vstep = gridDim.x;
        auto bounds = [&](unsigned v, int& ok, int& jj, int& b0, int& b1) {
            const unsigned tr = (v < vend ? cbl_xcd_slot(v, nwg) : 0u) * 4 + wave;
            ok = (v < vend && tr < n0) ? 1 : 0;
            const unsigned trc = ok ? tr : 0u;
            jj = order ? order[trc] : (int)trc; b0 = inv_start[trc]; b1 = inv_start[trc + 1];
        };
        auto pairs = [&](int ok, int jj, int b0, int b1, int& pp, float& x, float& y, float& z) {
            const int e = b0 + lane;
            pp = inv_src[(ok && e < b1) ? e : 0];                     // entry 0 always exists; lanes past the list are masked by the weight
            x = s[3 * jj]; y = s[3 * jj + 1]; z = s[3 * jj + 2];
        };
        int okA, jA, s0A, s1A, okB, jB, s0B, s1B, pB; float xB, yB, zB;
        bounds(blockIdx.x, okB, jB, s0B, s1B);
        pairs(okB, jB, s0B, s1B, pB, xB, yB, zB);
        bounds(blockIdx.x + vstep, okA, jA, s0A, s1A);
        for (unsigned v = blockIdx.x; v < vend; v += vstep) {
            const int ok = okB, jv = jB, s0v = s0B, s1v = s1B, p0 = pB; const float xj = xB, yj = yB, zj = zB;      // this trip's target
            okB = okA; jB = jA; s0B = s0A; s1B = s1A;
            pairs(okB, jB, s0B, s1B, pB, xB, yB, zB);                // pairs + coordinates of the next target
            bounds(v + 2 * vstep, okA, jA, s0A, s1A);                // bounds of the one after
            if (!__builtin_amdgcn_readfirstlane(ok)) continue;
            const int j = __builtin_amdgcn_readfirstlane(jv), s0 = __builtin_amdgcn_readfirstlane(s0v), s1 = __builtin_amdgcn_readfirstlane(s1v);
            f32x4 acc[4];
#pragma unroll
            for (int t = 0; t < 4; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int eb = s0; eb < s1; eb += 64) {
                // lane e holds pair eb + e: its query point and the offset of the target from it (entries past the end: clamped, weight 0 below)
                const int e = eb + lane;
                const int pi = (int)cbl_fastdiv((unsigned)(eb == s0 ? p0 : inv_src[e < s1 ? e : s0]), dvK);     // the first 64 pairs were prefetched
                const float3 qq = *reinterpret_cast<const float3*>(q + 3 * (size_t)pi);          // one 12-byte load per lane
                const float rx = xj - qq.x, ry = yj - qq.y, rz = zj - qq.z;                      // neighbour - centre (:681-684)
                const int cnt = min(64, s1 - eb);
                // one LDS round instead of four lane exchanges per step: lane e parks pair e, the step's lanes read the pair they multiply
                __builtin_amdgcn_wave_barrier();
                pairs_lds[wave][lane] = make_float4(__int_as_float(pi), rx, ry, rz);
                __builtin_amdgcn_wave_barrier();
                // Sixteen pairs (four MFMA steps) per batch: the four gradient rows of a lane are requested together, behind one round of
                // lane exchanges, and the weights are computed while they travel.  (One step at a time — exchange, wait, row, wait, multiply —
                // was a chain of ~1100 clocks per step, four to five per target, at four waves per SIMD: 46 us.)  Steps past the end of the
                // list re-read its last pair and multiply by zero.
                for (int sb = 0; sb < cnt; sb += 16) {
                    float4 g[4], pr[4]; float a[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) pr[u] = pairs_lds[wave][min(sb + 4 * u + kq, cnt - 1)];    // this lane's pair of step u
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        g[u] = *reinterpret_cast<const float4*>(go + (size_t)((unsigned)__float_as_int(pr[u].x) * (unsigned)C + (unsigned)cb));   // B operand: 16 lanes = one 256 B row segment per pair (32-bit element index: n C < 2^32 is checked by the launcher)
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int src = sb + 4 * u + kq;
                        const float dx = pr[u].y - kx, dy = pr[u].z - ky, dz = pr[u].w - kz;
                        const float sq = (dx * dx + dy * dy) + dz * dz;                          // :688
                        float w = influence ? fmaxf(1.0f - __builtin_amdgcn_sqrtf(sq) * inv_extent, 0.0f) : 1.0f;       // :697 / :693 (as the forward kernel)
                        if (CLOSEST) {                                                           // argmin over kernel points, first minimum (:705-708)
                            float bs = kp_ok ? sq : INFINITY; int bi = kp;
#pragma unroll
                            for (int sft = 8; sft >= 1; sft >>= 1) {
                                const float os = __shfl_xor(bs, sft, 16); const int oi = __shfl_xor(bi, sft, 16);
                                if (os < bs || (os == bs && oi < bi)) { bs = os; bi = oi; }
                            }
                            w = (bi != kp) ? 0.f : w;
                        }
                        a[u] = (kp_ok && src < cnt) ? w : 0.f;                                   // pairs past the end of the list contribute nothing
                    }
                    // every step is multiplied, also one past the end of the list (zero weights): a branch around the matrix instructions makes the
                    // compiler carry the sixteen accumulators through vector registers (32 moves per step)
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], g[u].x, acc[0], 0, 0, 0);          // S += W^T G
                        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], g[u].y, acc[1], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], g[u].z, acc[2], 0, 0, 0);
                        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], g[u].w, acc[3], 0, 0, 0);
                    }
                }
            }
            // the two contractions of S, once per target: tile t, row r of this lane <-> kernel point 4*kq + r, channel cb + t
            if (GF) {
                float res[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float4 kv = kw_lds[kq * 4 + r][kp];
                    res[0] = fmaf(kv.x, acc[0][r], res[0]); res[1] = fmaf(kv.y, acc[1][r], res[1]);
                    res[2] = fmaf(kv.z, acc[2][r], res[2]); res[3] = fmaf(kv.w, acc[3][r], res[3]);
                }
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    res[t] += __shfl_xor(res[t], 16);
                    res[t] += __shfl_xor(res[t], 32);
                }
                if (lane < 16 && cok) *reinterpret_cast<float4*>(gf + (size_t)j * C + cb) = make_float4(res[0], res[1], res[2], res[3]);
            }
            if (GKW) {
                const float4 fj = *reinterpret_cast<const float4*>(f + (size_t)j * C + cb);
                const float fv[4] = {fj.x, fj.y, fj.z, fj.w};
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) gk[t][r] = fmaf(fv[t], acc[t][r], gk[t][r]);
            }
        }
        if (GKW) {
            // every (kernel point, channel) of the chunk is held by exactly one lane of a wave: the four waves of the workgroup through LDS,
            // then one partial row block per workgroup
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) red[wave][kq * 4 + r][4 * kp + t] = gk[t][r];
            __syncthreads();
            for (int e = threadIdx.x; e < 16 * 64; e += KB_NB) {
                const int row = e >> 6, cc = e & 63;
                const float sum = (red[0][row][cc] + red[1][row][cc]) + (red[2][row][cc] + red[3][row][cc]);
                if (row < KP && c0 + cc < C) partial[((size_t)blockIdx.x * KP + row) * C + c0 + cc] = sum;
            }
        }
    }
}

// grad_kw[e] = sum over the workgroups' partials, in a fixed order (deterministic): 16 threads per element, each a 16th of the workgroups
__global__ __launch_bounds__(KB_NB) void kpconv_gkw_reduce_kernel(int nblk, int total, const float* __restrict__ partial, float* __restrict__ gkw)
{
    __shared__ float part[16][16];
    const int el = threadIdx.x & 15, pt = threadIdx.x >> 4;          // 16 elements per workgroup x 16 parts
    const int e = blockIdx.x * 16 + el;
    const int per = (nblk + 15) / 16, b0 = pt * per, b1 = min(nblk, b0 + per);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < total) {
        int b = b0;
        for (; b + 4 <= b1; b += 4) {
            a0 += partial[(size_t)b * total + e]; a1 += partial[(size_t)(b + 1) * total + e];
            a2 += partial[(size_t)(b + 2) * total + e]; a3 += partial[(size_t)(b + 3) * total + e];
        }
        for (; b < b1; b++) a0 += partial[(size_t)b * total + e];
    }
    part[pt][el] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (pt == 0 && e < total) {
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) sum += part[k][el];
        gkw[e] = sum;
    }
}

constexpr unsigned KB_MAX_GRID = 2048;                               // upper bound of the persistent launch (sizes the partial rows)

unsigned kb_grid(int n0)
{
    const unsigned g = cbl_round_up8(cbl_div_up(n0, 4));             // 4 target rows per workgroup and trip
    return g > KB_MAX_GRID ? KB_MAX_GRID : g;
}

// persistent waves: exactly as many workgroups as are resident at once (a few more would run as a second, mostly idle round)
constexpr int KB_MAX_DEVICES = 16;
template <class F> unsigned kb_resident(F kernel, int (&cache)[KB_MAX_DEVICES])
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= KB_MAX_DEVICES) dev = 0;       // one process per GPU: normally device 0 of its visible set
    if (!cache[dev]) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kernel), KB_NB, 0) != hipSuccess || per_cu <= 0) per_cu = 3;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cache[dev] = per_cu * cus;
    }
    return (unsigned)cache[dev];
}

}  // namespace

CBL_EXPORT size_t cbl_kpconv_backward_csr_workspace_bytes(int n0, int C, int KP)
{
    if (n0 <= 0 || C <= 0 || KP <= 0) return 0;
    return sizeof(float) * (size_t)kb_grid(n0) * (size_t)KP * (size_t)C;
}

CBL_EXPORT int cbl_kpconv_backward_csr(int n, int n0, int K, int C, int KP, const float* query_points, const float* support_points, const float* features,
                                       const float* kernel_points, const float* kernel_weights, float extent, int influence, int closest,
                                       const float* grad_out, const int* order_dst, const int* inv_start, const int* inv_src,
                                       float* grad_features, float* grad_kernel_weights, void* workspace, size_t workspace_bytes, void* stream)
{
    if (n < 0 || n0 < 0 || K <= 0 || C <= 0 || KP <= 0 || KP > 16 || !(extent > 0.f) || influence < 0 || influence > 1) return CBL_ERR_BAD_ARG;
    if (n0 == 0) return CBL_OK;
    if (!query_points || !support_points || !features || !kernel_points || !kernel_weights || !grad_out || !inv_start || !inv_src) return CBL_ERR_BAD_ARG;
    if (C % 4 || !cbl_host_aligned16(features) || !cbl_host_aligned16(grad_out) || !cbl_host_aligned16(kernel_weights) ||
        (grad_features && !cbl_host_aligned16(grad_features))) return CBL_ERR_UNSUPPORTED;
    if (!grad_features && !grad_kernel_weights) return CBL_OK;
    if ((unsigned long long)n * (unsigned long long)C >= (1ull << 32)) return CBL_ERR_UNSUPPORTED;     // 32-bit element index of a gradient row
    hipStream_t st = cbl_stream(stream);
    unsigned g = kb_grid(n0);
    float* partial = reinterpret_cast<float*>(workspace);
    if (grad_kernel_weights && (!partial || workspace_bytes < cbl_kpconv_backward_csr_workspace_bytes(n0, C, KP))) return CBL_ERR_WORKSPACE;
    const CblFastDiv dv = cbl_fastdiv_make((unsigned)K);
    static int resident[2][2][2][KB_MAX_DEVICES] = {};
#define CBL_KB(GF_, GKW_, CL_) { const unsigned res = cbl_round_up8(kb_resident(&kpconv_bwd_csr_kernel<GF_, GKW_, CL_>, resident[GF_][GKW_][CL_])); if (g > res) g = res;  \
        hipLaunchKernelGGL((kpconv_bwd_csr_kernel<GF_, GKW_, CL_>), dim3(g), dim3(KB_NB), 0, st, (unsigned)n0, C, KP, dv, query_points, support_points, \
        features, kernel_points, kernel_weights, extent, influence, grad_out, order_dst, inv_start, inv_src, grad_features, partial); }
    if (closest) {
        if (grad_features && grad_kernel_weights) { CBL_KB(true, true, true) } else if (grad_features) { CBL_KB(true, false, true) } else { CBL_KB(false, true, true) }
    } else {
        if (grad_features && grad_kernel_weights) { CBL_KB(true, true, false) } else if (grad_features) { CBL_KB(true, false, false) } else { CBL_KB(false, true, false) }
    }
#undef CBL_KB
    if (grad_kernel_weights)
        hipLaunchKernelGGL(kpconv_gkw_reduce_kernel, dim3(cbl_div_up(KP * C, 16)), dim3(KB_NB), 0, st, (int)g, KP * C, partial, grad_kernel_weights);
    return cbl_status();
}
