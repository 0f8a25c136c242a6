// a15 backward without atomics: KPConv's gradients as a gather over the transposed neighbour table.
//   PseudoGrid math   /root/reference/tensorflow/models/local_aggregation_operators.py:681-728
//     out[i,c] = sum_kp kw[kp,c] * sum_k w[i,kp,k] * f[nbr(i,k), c],   w = influence of kernel point kp on neighbour k of point i
//   d out / d f    : grad_f[j,c]   = sum over the pairs p = (i,k) with nbr(i,k) = j of  go[i,c] * sum_kp w[p,kp] kw[kp,c]
//   d out / d kw   : grad_kw[kp,c] = sum over ALL pairs of                              w[p,kp] * f[j,c] * go[i,c]
// Round 1 walked the query points and scattered d out / d f with one float atomic per (pair, channel): 42 M atomics on the memory side
// of the fabric, 298 us at N = 40960, K = 16, C = 64.  Here one wave owns a TARGET row j and walks its pairs (cbl_neighbor_transpose),
// four at a time: lane = (pair slot, 4 channels).  The 16 lanes of a pair slot each compute ONE influence weight (lane q: kernel point q)
// and the weights travel across the 16 lanes by DPP row rotation — step r hands lane q the weight of kernel point src(q, r); the kernel
// weights each lane needs at step r (row src(q, r), its own 4 channels) are loaded once per wave in that rotated order.  Both gradients
// come out of the same 16 steps: h += w * kw (then grad_f += go * h) and grad_kw's accumulator r += w * (f_j * go_i).  grad_f is written
// with plain 16-byte stores; grad_kw is reduced over the lanes of a wave, the waves of a workgroup (LDS) and the workgroups
// (per-workgroup partial rows + one small reduction kernel): deterministic, no atomics anywhere.
#include "cbl_common.h"
#include "wave_ops.h"

namespace {

constexpr int KB_NB = 256;

template <int R> __device__ __forceinline__ float rot_f(float v) { return R == 0 ? v : dpp_mov_f<0x120 + (R == 0 ? 1 : R), 0xf>(v); }
template <int R> __device__ __forceinline__ int rot_i(int v) { return R == 0 ? v : dpp_mov_i<0x120 + (R == 0 ? 1 : R), 0xf>(v); }

template <int R, bool GKW> struct RotSteps {
    // steps R .. 15 of the rotation: h += w_r * kwrot[r];  gk[r] += w_r * m
    static __device__ __forceinline__ void run(float w, const float4 (&kwrot)[16], float4 (&gk)[16], const float4& m, float4& h)
    {
        const float wr = rot_f<R>(w);
        h.x = fmaf(wr, kwrot[R].x, h.x); h.y = fmaf(wr, kwrot[R].y, h.y); h.z = fmaf(wr, kwrot[R].z, h.z); h.w = fmaf(wr, kwrot[R].w, h.w);
        if (GKW) { gk[R].x = fmaf(wr, m.x, gk[R].x); gk[R].y = fmaf(wr, m.y, gk[R].y); gk[R].z = fmaf(wr, m.z, gk[R].z); gk[R].w = fmaf(wr, m.w, gk[R].w); }
        RotSteps<R + 1, GKW>::run(w, kwrot, gk, m, h);
    }
    static __device__ __forceinline__ void load(int ql, int KP, int C, int cb, bool cok, const float* __restrict__ kw, float4 (&kwrot)[16], int (&srck)[16])
    {
        const int src = rot_i<R>(ql);                                // the lane whose weight arrives at step R = its kernel point
        srck[R] = src;
        kwrot[R] = (src < KP && cok) ? *reinterpret_cast<const float4*>(kw + (size_t)src * C + cb) : make_float4(0.f, 0.f, 0.f, 0.f);
        RotSteps<R + 1, GKW>::load(ql, KP, C, cb, cok, kw, kwrot, srck);
    }
};
template <bool GKW> struct RotSteps<16, GKW> {
    static __device__ __forceinline__ void run(float, const float4 (&)[16], float4 (&)[16], const float4&, float4&) {}
    static __device__ __forceinline__ void load(int, int, int, int, bool, const float* __restrict__, float4 (&)[16], int (&)[16]) {}
};

// one wave per target row; C % 4 == 0, rows 16-byte aligned, KP <= 16.  partial: (gridDim.x, KP, C) per-workgroup sums of grad_kw.
template <bool GF, bool GKW>
__global__ __launch_bounds__(KB_NB) void kpconv_bwd_csr_kernel(unsigned n0, int C, int KP, CblFastDiv dvK, const float* __restrict__ q, const float* __restrict__ s,
                                                               const float* __restrict__ f, const float* __restrict__ kpts, const float* __restrict__ kw,
                                                               float extent, int influence, int closest, const float* __restrict__ go,
                                                               const int* __restrict__ order, const int* __restrict__ inv_start, const int* __restrict__ inv_src,
                                                               float* __restrict__ gf, float* __restrict__ partial)
{
    __shared__ float red[KB_NB / 64][16][64];                        // grad_kw of the four waves: [wave][kernel point][channel of the chunk]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = lane >> 4, ql = lane & 15;
    const bool kp_ok = ql < KP;
    const float kx = kp_ok ? kpts[3 * ql] : 0.f, ky = kp_ok ? kpts[3 * ql + 1] : 0.f, kz = kp_ok ? kpts[3 * ql + 2] : 0.f;
    const float inv_extent = 1.0f / extent;
    const unsigned nwg = (n0 + 3) >> 2;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int cb = c0 + 4 * ql;
        const bool cok = cb < C;
        float4 kwrot[16], gk[16]; int srck[16];
        RotSteps<0, GKW>::load(ql, KP, C, cb, cok, kw, kwrot, srck);
#pragma unroll
        for (int r = 0; r < 16; r++) gk[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (unsigned v = blockIdx.x; v < 8 * cbl_xcd_per(nwg); v += gridDim.x) {
            const unsigned tr = cbl_xcd_slot(v, nwg) * 4 + wave;
            if (tr >= n0) continue;
            const int j = order ? order[tr] : (int)tr;
            const int s0 = inv_start[tr], s1 = inv_start[tr + 1];
            const float xj = s[3 * j], yj = s[3 * j + 1], zj = s[3 * j + 2];
            const float4 fj = (GKW && cok) ? *reinterpret_cast<const float4*>(f + (size_t)j * C + cb) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int base = s0; base < s1; base += 8) {              // two groups of four pairs: their loads are issued together
                int pi[2]; bool ok[2]; float rx[2], ry[2], rz[2]; float4 g[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int e = base + 4 * u + slot;
                    ok[u] = e < s1;
                    pi[u] = (int)cbl_fastdiv((unsigned)(ok[u] ? inv_src[e] : 0), dvK);      // query point of the pair
                }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    rx[u] = xj - q[3 * pi[u]]; ry[u] = yj - q[3 * pi[u] + 1]; rz[u] = zj - q[3 * pi[u] + 2];      // neighbour - centre (:681-684)
                    g[u] = (ok[u] && cok) ? *reinterpret_cast<const float4*>(go + (size_t)pi[u] * C + cb) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    if (u == 1 && base + 4 >= s1) break;             // wave-uniform
                    const float dx = rx[u] - kx, dy = ry[u] - ky, dz = rz[u] - kz;
                    const float sq = (dx * dx + dy * dy) + dz * dz;                          // :688
                    float w = influence ? fmaxf(1.0f - __builtin_amdgcn_sqrtf(sq) * inv_extent, 0.0f) : 1.0f;   // :697 / :693 (as the forward kernel)
                    if (closest) {                                                           // argmin over kernel points, first minimum (:705-708)
                        float bs = kp_ok ? sq : INFINITY; int bi = ql;
#pragma unroll
                        for (int sft = 8; sft >= 1; sft >>= 1) {
                            const float os = __shfl_xor(bs, sft, 16); const int oi = __shfl_xor(bi, sft, 16);
                            if (os < bs || (os == bs && oi < bi)) { bs = os; bi = oi; }
                        }
                        if (bi != ql) w = 0.f;
                    }
                    w = (kp_ok && ok[u]) ? w : 0.f;
                    const float4 m = make_float4(fj.x * g[u].x, fj.y * g[u].y, fj.z * g[u].z, fj.w * g[u].w);
                    float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
                    RotSteps<0, GKW>::run(w, kwrot, gk, m, h);
                    if (GF) { acc.x = fmaf(g[u].x, h.x, acc.x); acc.y = fmaf(g[u].y, h.y, acc.y); acc.z = fmaf(g[u].z, h.z, acc.z); acc.w = fmaf(g[u].w, h.w, acc.w); }
                }
            }
            if (GF) {
                acc.x += __shfl_xor(acc.x, 16); acc.y += __shfl_xor(acc.y, 16); acc.z += __shfl_xor(acc.z, 16); acc.w += __shfl_xor(acc.w, 16);
                acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32); acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
                if (slot == 0 && cok) *reinterpret_cast<float4*>(gf + (size_t)j * C + cb) = acc;
            }
        }
        if (GKW) {
            // the four pair slots of a wave, then the four waves of the workgroup, then one partial row block per workgroup
#pragma unroll
            for (int r = 0; r < 16; r++) {
                float4 t = gk[r];
                t.x += __shfl_xor(t.x, 16); t.y += __shfl_xor(t.y, 16); t.z += __shfl_xor(t.z, 16); t.w += __shfl_xor(t.w, 16);
                t.x += __shfl_xor(t.x, 32); t.y += __shfl_xor(t.y, 32); t.z += __shfl_xor(t.z, 32); t.w += __shfl_xor(t.w, 32);
                if (slot == 0) *reinterpret_cast<float4*>(&red[wave][srck[r]][4 * ql]) = t;
            }
            __syncthreads();
            for (int e = threadIdx.x; e < 16 * 64; e += KB_NB) {
                const int kp = e >> 6, cc = e & 63;
                const float sum = (red[0][kp][cc] + red[1][kp][cc]) + (red[2][kp][cc] + red[3][kp][cc]);
                if (kp < KP && c0 + cc < C) partial[((size_t)blockIdx.x * KP + kp) * C + c0 + cc] = sum;
            }
            __syncthreads();
        }
    }
}

// grad_kw[e] = sum over the workgroups' partials, in workgroup order (deterministic)
__global__ __launch_bounds__(KB_NB) void kpconv_gkw_reduce_kernel(int nblk, int total, const float* __restrict__ partial, float* __restrict__ gkw)
{
    const int e = blockIdx.x * KB_NB + threadIdx.x;
    if (e >= total) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int b = 0;
    for (; b + 4 <= nblk; b += 4) {
        a0 += partial[(size_t)b * total + e]; a1 += partial[(size_t)(b + 1) * total + e];
        a2 += partial[(size_t)(b + 2) * total + e]; a3 += partial[(size_t)(b + 3) * total + e];
    }
    for (; b < nblk; b++) a0 += partial[(size_t)b * total + e];
    gkw[e] = (a0 + a1) + (a2 + a3);
}

unsigned kb_grid(int n0)
{
    unsigned g = cbl_round_up8(cbl_div_up(n0, 4));
    return g > 1024u ? 1024u : g;
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
        hipLaunchKernelGGL(kpconv_gkw_reduce_kernel, dim3(cbl_div_up(KP * C, KB_NB)), dim3(KB_NB), 0, st, (int)g, KP * C, partial, grad_kernel_weights);
    return cbl_status();
}
