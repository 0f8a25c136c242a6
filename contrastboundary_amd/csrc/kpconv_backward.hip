// a15 backward without atomics: KPConv's gradients as a gather over the transposed neighbour table.
//   PseudoGrid math   /root/reference/tensorflow/models/local_aggregation_operators.py:681-728
//     out[i,c] = sum_kp kw[kp,c] * sum_k w[i,kp,k] * f[nbr(i,k), c],   w = influence of kernel point kp on neighbour k of point i
//   d out / d f    : grad_f[j,c]   = sum over the pairs p = (i,k) with nbr(i,k) = j of  go[i,c] * sum_kp w[p,kp] kw[kp,c]
//   d out / d kw   : grad_kw[kp,c] = sum over ALL pairs of                              w[p,kp] * f[j,c] * go[i,c]
// Round 1 walked the query points and scattered d out / d f with one float atomic per (pair, channel): 42 M atomics on the memory side
// of the fabric, 298 us at N = 40960, K = 16, C = 64.  Here 16 lanes own a TARGET row j (lane = 4 channels) and walk its pairs
// (cbl_neighbor_transpose).  For every pair the 16 lanes each compute ONE influence weight (lane q: kernel point q)
// and the weights travel across the 16 lanes by DPP row rotation — step r hands lane q the weight of kernel point src(q, r) — into the
// per-target sum S_j[kp,c] = sum_p w[p,kp] go[i_p,c], from which both gradients follow once per target.  grad_f is written
// with plain 16-byte stores; grad_kw is reduced over the lanes of a wave, the waves of a workgroup (LDS) and the workgroups
// (per-workgroup partial rows + one small reduction kernel): deterministic, no atomics anywhere.
#include "cbl_common.h"
#include "wave_ops.h"

namespace {

constexpr int KB_NB = 256;

template <int R> __device__ __forceinline__ float rot_f(float v) { return R == 0 ? v : dpp_mov_f<0x120 + (R == 0 ? 1 : R), 0xf>(v); }
template <int R> __device__ __forceinline__ int rot_i(int v) { return R == 0 ? v : dpp_mov_i<0x120 + (R == 0 ? 1 : R), 0xf>(v); }

// Both gradients go through ONE per-target sum:  S_j[kp,c] = sum over the pairs p of target j of  w[p,kp] * go[i_p,c]
//   grad_f[j,c]   = sum_kp kw[kp,c] * S_j[kp,c]          grad_kw[kp,c] += f[j,c] * S_j[kp,c]
// so a pair costs KP multiply-adds per channel (not 2 KP), and the two contractions with kw / f_j are paid once per target.
template <int R> struct RotSteps {
    // steps R .. 15 of the rotation: S[r] += w_r * g   (w_r = the weight of kernel point src(lane, r), handed over by DPP row rotation)
    static __device__ __forceinline__ void accumulate(float w, const float4& g, float4 (&S)[16])
    {
        const float wr = rot_f<R>(w);
        S[R].x = fmaf(wr, g.x, S[R].x); S[R].y = fmaf(wr, g.y, S[R].y); S[R].z = fmaf(wr, g.z, S[R].z); S[R].w = fmaf(wr, g.w, S[R].w);
        RotSteps<R + 1>::accumulate(w, g, S);
    }
    static __device__ __forceinline__ void sources(int ql, int (&srck)[16])
    {
        srck[R] = rot_i<R>(ql);                                      // the lane whose weight arrives at step R = its kernel point
        RotSteps<R + 1>::sources(ql, srck);
    }
};
template <> struct RotSteps<16> {
    static __device__ __forceinline__ void accumulate(float, const float4&, float4 (&)[16]) {}
    static __device__ __forceinline__ void sources(int, int (&)[16]) {}
};

// 16 lanes per target row (four targets per wave, sixteen per workgroup); C % 4 == 0, rows 16-byte aligned, KP <= 16.
// A group walks its target's pairs sixteen at a time: lane e of the group fetches pair e (its query point and the offset to it) in two
// round trips for all sixteen, then the pairs are taken one by one — broadcast inside the group (ds_bpermute, one pair ahead), one
// influence weight per lane, the 16 rotation steps into S.  Four independent targets per wave and four gradient rows in flight per group
// cover the L2 round trips; nothing is reduced across lanes for grad_f (each lane owns 4 channels of its target).  Every load is
// unconditional with a clamped address and masked afterwards, and every loop has a scalar trip count: exec-masked blocks and
// vector-controlled loops made the compiler keep three copies of the 128 accumulators.
// partial: (gridDim.x, KP, C) per-workgroup sums of grad_kw.
template <bool GF, bool GKW>
__global__ __launch_bounds__(KB_NB) void kpconv_bwd_csr_kernel(unsigned n0, int C, int KP, CblFastDiv dvK, const float* __restrict__ q, const float* __restrict__ s,
                                                               const float* __restrict__ f, const float* __restrict__ kpts, const float* __restrict__ kw,
                                                               float extent, int influence, int closest, const float* __restrict__ go,
                                                               const int* __restrict__ order, const int* __restrict__ inv_start, const int* __restrict__ inv_src,
                                                               float* __restrict__ gf, float* __restrict__ partial)
{
    __shared__ float red[KB_NB / 64][16][64];                        // grad_kw of the four waves: [wave][kernel point][channel of the chunk]
    __shared__ float kw_s[16][64];                                   // kernel weights of the channel chunk (rows >= KP and channels >= C: 0)
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = lane >> 4, ql = lane & 15;
    const bool kp_ok = ql < KP;
    const float kx = kp_ok ? kpts[3 * ql] : 0.f, ky = kp_ok ? kpts[3 * ql + 1] : 0.f, kz = kp_ok ? kpts[3 * ql + 2] : 0.f;
    const float inv_extent = 1.0f / extent;
    const unsigned nwg = (n0 + 15) >> 4;
    int srck[16];
    RotSteps<0>::sources(ql, srck);
    for (int c0 = 0; c0 < C; c0 += 64) {
        const bool cok = c0 + 4 * ql < C;
        const int cb = cok ? c0 + 4 * ql : c0;                       // lanes beyond the last channel read the chunk's first four and contribute nothing
        __syncthreads();
        for (int e = threadIdx.x; e < 16 * 64; e += KB_NB) {
            const int kp = e >> 6, cc = e & 63;
            kw_s[kp][cc] = (kp < KP && c0 + cc < C) ? kw[(size_t)kp * C + c0 + cc] : 0.f;
        }
        __syncthreads();
        float4 gk[16];
#pragma unroll
        for (int r = 0; r < 16; r++) gk[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (unsigned v = blockIdx.x; v < 8 * cbl_xcd_per(nwg); v += gridDim.x) {
            const unsigned tr = cbl_xcd_slot(v, nwg) * 16 + wave * 4 + grp;
            const bool tok = tr < n0;
            const int j = tok ? (order ? order[tr] : (int)tr) : 0;
            const int s0 = tok ? inv_start[tr] : 0, s1 = tok ? inv_start[tr + 1] : 0;
            const float xj = s[3 * j], yj = s[3 * j + 1], zj = s[3 * j + 2];
            float4 S[16];
#pragma unroll
            for (int r = 0; r < 16; r++) S[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            // ONE loop over the pairs of the longest target of the wave (a scalar trip count; a second loop level around the 64 accumulators cost
            // ~50 registers): every 16 trips lane e of a group fetches pair e of the next sixteen (its query point and the offset to it);
            // four gradient rows in flight per group, the offset of the next pair broadcast one trip ahead
            const int L = s1 - s0;
            int most = L;
            most = max(most, __shfl_xor(most, 16)); most = max(most, __shfl_xor(most, 32));
            most = __builtin_amdgcn_readfirstlane(most);
            int pi = 0; float rx = 0.f, ry = 0.f, rz = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
            float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, g2 = g0, g3 = g0;
            auto row = [&](int t) -> float4 {
                const int pit = __shfl(pi, t & 15, 16);
                return *reinterpret_cast<const float4*>(go + (size_t)pit * C + cb);         // pairs beyond L: some valid row, weight 0 below
            };
#pragma unroll 1
            for (int t = 0; t < most; t++) {
                if ((t & 15) == 0) {                                 // wave-uniform
                    const int e = s0 + t + ql;
                    pi = (int)cbl_fastdiv((unsigned)inv_src[e < s1 ? e : 0], dvK);             // query point of pair e (entry 0 always exists)
                    rx = xj - q[3 * pi]; ry = yj - q[3 * pi + 1]; rz = zj - q[3 * pi + 2];     // neighbour - centre (:681-684)
                    g0 = row(0); g1 = row(1); g2 = row(2); g3 = row(3);
                    nx = __shfl(rx, 0, 16); ny = __shfl(ry, 0, 16); nz = __shfl(rz, 0, 16);
                }
                const float4 g = g0;
                const float dx = nx - kx, dy = ny - ky, dz = nz - kz;
                g0 = g1; g1 = g2; g2 = g3; g3 = row(t + 4);          // (rows past the sixteen are refetched at the next boundary)
                nx = __shfl(rx, (t + 1) & 15, 16); ny = __shfl(ry, (t + 1) & 15, 16); nz = __shfl(rz, (t + 1) & 15, 16);
                const float sq = (dx * dx + dy * dy) + dz * dz;                                  // :688
                float w = influence ? fmaxf(1.0f - __builtin_amdgcn_sqrtf(sq) * inv_extent, 0.0f) : 1.0f;           // :697 / :693 (as the forward kernel)
                if (closest) {                                                                   // argmin over kernel points, first minimum (:705-708)
                    float bs = kp_ok ? sq : INFINITY; int bi = ql;
#pragma unroll
                    for (int sft = 8; sft >= 1; sft >>= 1) {
                        const float os = __shfl_xor(bs, sft, 16); const int oi = __shfl_xor(bi, sft, 16);
                        if (os < bs || (os == bs && oi < bi)) { bs = os; bi = oi; }
                    }
                    if (bi != ql) w = 0.f;
                }
                w = (kp_ok && t < L) ? w : 0.f;                                                 // also silences the rows fetched beyond L
                RotSteps<0>::accumulate(w, g, S);
            }
            // the two contractions of S, once per target: grad_f with the kernel weights (LDS), grad_kw's accumulators with the target's features
            const float4 fj = GKW ? *reinterpret_cast<const float4*>(f + (size_t)j * C + cb) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                if (GF) {
                    const float4 kv = *reinterpret_cast<const float4*>(&kw_s[srck[r]][4 * ql]);
                    acc.x = fmaf(kv.x, S[r].x, acc.x); acc.y = fmaf(kv.y, S[r].y, acc.y); acc.z = fmaf(kv.z, S[r].z, acc.z); acc.w = fmaf(kv.w, S[r].w, acc.w);
                }
                if (GKW) { gk[r].x = fmaf(fj.x, S[r].x, gk[r].x); gk[r].y = fmaf(fj.y, S[r].y, gk[r].y); gk[r].z = fmaf(fj.z, S[r].z, gk[r].z); gk[r].w = fmaf(fj.w, S[r].w, gk[r].w); }
            }
            if (GF && tok && cok) *reinterpret_cast<float4*>(gf + (size_t)j * C + cb) = acc;
        }
        if (GKW) {
            // the four groups of a wave, then the four waves of the workgroup, then one partial row block per workgroup
#pragma unroll
            for (int r = 0; r < 16; r++) {
                float4 t = gk[r];
                t.x += __shfl_xor(t.x, 16); t.y += __shfl_xor(t.y, 16); t.z += __shfl_xor(t.z, 16); t.w += __shfl_xor(t.w, 16);
                t.x += __shfl_xor(t.x, 32); t.y += __shfl_xor(t.y, 32); t.z += __shfl_xor(t.z, 32); t.w += __shfl_xor(t.w, 32);
                if (grp == 0) *reinterpret_cast<float4*>(&red[wave][srck[r]][4 * ql]) = t;
            }
            __syncthreads();
            for (int e = threadIdx.x; e < 16 * 64; e += KB_NB) {
                const int kp = e >> 6, cc = e & 63;
                const float sum = (red[0][kp][cc] + red[1][kp][cc]) + (red[2][kp][cc] + red[3][kp][cc]);
                if (kp < KP && c0 + cc < C) partial[((size_t)blockIdx.x * KP + kp) * C + c0 + cc] = sum;
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

unsigned kb_grid(int n0)
{
    unsigned g = cbl_round_up8(cbl_div_up(n0, 16));                    // 16 target rows per workgroup and trip
    return g > 768u ? 768u : g;
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
    hipStream_t st = cbl_stream(stream);
    const unsigned g = kb_grid(n0);
    float* partial = reinterpret_cast<float*>(workspace);
    if (grad_kernel_weights && (!partial || workspace_bytes < cbl_kpconv_backward_csr_workspace_bytes(n0, C, KP))) return CBL_ERR_WORKSPACE;
    const CblFastDiv dv = cbl_fastdiv_make((unsigned)K);
#define CBL_KB(GF_, GKW_) hipLaunchKernelGGL((kpconv_bwd_csr_kernel<GF_, GKW_>), dim3(g), dim3(KB_NB), 0, st, (unsigned)n0, C, KP, dv, query_points, support_points, \
        features, kernel_points, kernel_weights, extent, influence, closest, grad_out, order_dst, inv_start, inv_src, grad_features, partial)
    if (grad_features && grad_kernel_weights) CBL_KB(true, true);
    else if (grad_features) CBL_KB(true, false);
    else CBL_KB(false, true);
#undef CBL_KB
    if (grad_kernel_weights)
        hipLaunchKernelGGL(kpconv_gkw_reduce_kernel, dim3(cbl_div_up(KP * C, 16)), dim3(KB_NB), 0, st, (int)g, KP * C, partial, grad_kernel_weights);
    return cbl_status();
}
