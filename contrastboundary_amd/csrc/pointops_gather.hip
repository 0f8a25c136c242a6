// K3..K10 + F1/F4: neighbour-index gather / scatter family (all HBM-bound).
// Replaces /root/reference/pytorch/lib/pointops/src/{grouping,interpolation,subtraction,aggregation}/*_cuda_kernel.cu
// and the torch-op chain of queryandgroup (pointops.py:79-100).
//
// MI355X mapping: the reference runs one thread per output SCALAR and re-derives (row, k, channel) by
// division, re-loading idx once per scalar.  Here a lane owns 4 consecutive channels (16 B, one
// global_load_dwordx4 / global_store_dwordx4), so a wave moves whole 64..256 B feature rows with
// fully coalesced 1 KiB stores, idx is loaded once per 4 channels, and the grid is a fixed
// 256 CU x 16 grid-stride launch.  Rows whose channel count is not a multiple of 4 (xyz: c = 3, the
// 3+c concat) take the scalar path with the same values.  Accumulating ops (+=) read the caller's
// pre-zeroed output, like the reference, and sum neighbours in index order without FMA contraction
// so forward results are bit-identical to the CPU oracle; scatter-add backward passes use hardware
// fp32 atomics (order-dependent rounding only, as in the reference).
#include "cbl_common.h"
#include <stdlib.h>

namespace {

constexpr int GB = 256;  // threads per block

__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

// ---------------------------------------------------------------- K3 grouping forward
// out[r, :] = in[idx[r], :]   r over m*nsample rows                 grouping_cuda_kernel.cu:5-14
__global__ __launch_bounds__(GB) void grouping_fwd_v4(long long rows, int c4, const float4* __restrict__ in,
                                                      const int* __restrict__ idx, float4* __restrict__ out)
{
    const long long total = rows * c4;
    for (long long e = (long long)blockIdx.x * GB + threadIdx.x; e < total; e += (long long)gridDim.x * GB) {
        const long long r = e / c4; const int ch = (int)(e - r * c4);
        // the (m,K,c) output is written once and read by a later kernel: streaming (nt) store keeps the gathered source rows in L2
        using v4 = __attribute__((ext_vector_type(4))) float;
        const float4 v = in[(long long)idx[r] * c4 + ch];
        __builtin_nontemporal_store(v4{v.x, v.y, v.z, v.w}, reinterpret_cast<v4*>(out + e));
    }
}
// the same with a processing order over the m query points (cbl_common.h): 256-lane chunks of the (sequence slot, neighbour, part) space
// are dealt to the XCDs in contiguous eighths, slot t stands for point order[t]
__global__ __launch_bounds__(GB) void grouping_fwd_v4_ordered(unsigned m, int ns, int c4, const float4* __restrict__ in, const int* __restrict__ idx,
                                                              const int* __restrict__ order, float4* __restrict__ out)
{
    const unsigned per_pt = (unsigned)ns * (unsigned)c4;
    const unsigned long long total = (unsigned long long)m * per_pt;
    const unsigned nch = (unsigned)((total + GB - 1) / GB);
    for (unsigned v = blockIdx.x; v < 8 * cbl_xcd_per(nch); v += gridDim.x) {
        const unsigned long long e = (unsigned long long)cbl_xcd_slot(v, nch) * GB + threadIdx.x;
        if (e >= total) continue;
        const unsigned t = (unsigned)(e / per_pt), j = (unsigned)(e - (unsigned long long)t * per_pt);
        const unsigned k = j / (unsigned)c4, ch = j - k * (unsigned)c4;
        const size_t r = (size_t)order[t] * ns + k;
        using v4 = __attribute__((ext_vector_type(4))) float;
        const float4 val = in[(size_t)idx[r] * c4 + ch];
        __builtin_nontemporal_store(v4{val.x, val.y, val.z, val.w}, reinterpret_cast<v4*>(out + r * c4 + ch));
    }
}
__global__ __launch_bounds__(GB) void grouping_fwd_s(long long rows, int c, const float* __restrict__ in,
                                                     const int* __restrict__ idx, float* __restrict__ out)
{
    const long long total = rows * c;
    for (long long e = (long long)blockIdx.x * GB + threadIdx.x; e < total; e += (long long)gridDim.x * GB) {
        const long long r = e / c; const int ch = (int)(e - r * c);
        out[e] = in[(long long)idx[r] * c + ch];
    }
}

// ---------------------------------------------------------------- K4 grouping backward
// grad_in[idx[r], :] += grad_out[r, :]                              grouping_cuda_kernel.cu:16-25
__global__ __launch_bounds__(GB) void grouping_bwd(long long rows, int c, const float* __restrict__ go,
                                                   const int* __restrict__ idx, float* __restrict__ gi)
{
    const long long total = rows * c;
    for (long long e = (long long)blockIdx.x * GB + threadIdx.x; e < total; e += (long long)gridDim.x * GB) {
        const long long r = e / c; const int ch = (int)(e - r * c);
        atomic_add_f32(gi + (long long)idx[r] * c + ch, go[e]);
    }
}

// ---------------------------------------------------------------- K5 interpolation forward
// out[p, ch] += sum_i in[idx[p,i], ch] * w[p,i]   (i ascending)     interpolation_cuda_kernel.cu:5-18
template <int V>
__global__ __launch_bounds__(GB) void interp_fwd(int n, int cv, int k, const float* __restrict__ in,
                                                 const int* __restrict__ idx, const float* __restrict__ w, float* __restrict__ out)
{
    const long long total = (long long)n * cv;
    for (long long e = (long long)blockIdx.x * GB + threadIdx.x; e < total; e += (long long)gridDim.x * GB) {
        const long long p = e / cv; const int ch = (int)(e - p * cv);
        if (V == 4) {
            float4 acc = reinterpret_cast<float4*>(out)[e];
            for (int i = 0; i < k; i++) {
                const float wi = w[p * k + i];
                const float4 v = reinterpret_cast<const float4*>(in)[(long long)idx[p * k + i] * cv + ch];
                acc.x += v.x * wi; acc.y += v.y * wi; acc.z += v.z * wi; acc.w += v.w * wi;
            }
            reinterpret_cast<float4*>(out)[e] = acc;
        } else {
            float acc = out[e];
            for (int i = 0; i < k; i++) acc += in[(long long)idx[p * k + i] * cv + ch] * w[p * k + i];
            out[e] = acc;
        }
    }
}

// ---------------------------------------------------------------- K6 interpolation backward
// grad_in[idx[p,i], ch] += grad_out[p, ch] * w[p,i]                 interpolation_cuda_kernel.cu:20-33
__global__ __launch_bounds__(GB) void interp_bwd(int n, int c, int k, const float* __restrict__ go,
                                                 const int* __restrict__ idx, const float* __restrict__ w, float* __restrict__ gi)
{
    const long long total = (long long)n * c;
    for (long long e = (long long)blockIdx.x * GB + threadIdx.x; e < total; e += (long long)gridDim.x * GB) {
        const long long p = e / c; const int ch = (int)(e - p * c);
        const float g = go[e];
        for (int i = 0; i < k; i++) atomic_add_f32(gi + (long long)idx[p * k + i] * c + ch, g * w[p * k + i]);
    }
}

// ---------------------------------------------------------------- K7 subtraction forward
// out[p,s,:] = in1[p,:] - in2[idx[p,s],:]                           subtraction_cuda_kernel.cu:5-16
template <int V>
__global__ __launch_bounds__(GB) void sub_fwd(long long rows, int ns, int cv, const float* __restrict__ a,
                                              const float* __restrict__ b2, const int* __restrict__ idx, float* __restrict__ out)
{
    const long long total = rows * cv;
    for (long long e = (long long)blockIdx.x * GB + threadIdx.x; e < total; e += (long long)gridDim.x * GB) {
        const long long r = e / cv; const int ch = (int)(e - r * cv);
        const long long p = r / ns;
        if (V == 4) {
            const float4 x = reinterpret_cast<const float4*>(a)[p * cv + ch];
            const float4 y = reinterpret_cast<const float4*>(b2)[(long long)idx[r] * cv + ch];
            reinterpret_cast<float4*>(out)[e] = make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w);
        } else {
            out[e] = a[p * cv + ch] - b2[(long long)idx[r] * cv + ch];
        }
    }
}
// with a processing order over the points (cbl_common.h): chunks of the (sequence slot, neighbour, part) space dealt XCD-contiguously
__global__ __launch_bounds__(GB) void sub_fwd_v4_ordered(unsigned n, int ns, int cv, const float4* __restrict__ a, const float4* __restrict__ b2,
                                                         const int* __restrict__ idx, const int* __restrict__ order, float4* __restrict__ out)
{
    const unsigned per_pt = (unsigned)ns * (unsigned)cv;
    const unsigned long long total = (unsigned long long)n * per_pt;
    const unsigned nch = (unsigned)((total + GB - 1) / GB);
    for (unsigned v = blockIdx.x; v < 8 * cbl_xcd_per(nch); v += gridDim.x) {
        const unsigned long long e = (unsigned long long)cbl_xcd_slot(v, nch) * GB + threadIdx.x;
        if (e >= total) continue;
        const unsigned t = (unsigned)(e / per_pt), j = (unsigned)(e - (unsigned long long)t * per_pt);
        const unsigned k = j / (unsigned)cv, ch = j - k * (unsigned)cv;
        const size_t p = (size_t)order[t], r = p * ns + k;
        const float4 x = a[p * cv + ch];
        const float4 y = b2[(size_t)idx[r] * cv + ch];
        out[r * cv + ch] = make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w);
    }
}

// ---------------------------------------------------------------- K8 subtraction backward
// g1[p,:] += go[p,s,:] ; g2[idx[p,s],:] += -go[p,s,:]               subtraction_cuda_kernel.cu:18-30
// g1 is a per-row sum over s: done in registers by the lane that owns (p, ch) — no atomics, fixed order.
__global__ __launch_bounds__(GB) void sub_bwd(int n, int ns, int c, const int* __restrict__ idx,
                                              const float* __restrict__ go, float* __restrict__ g1, float* __restrict__ g2)
{
    const long long total = (long long)n * c;
    for (long long e = (long long)blockIdx.x * GB + threadIdx.x; e < total; e += (long long)gridDim.x * GB) {
        const long long p = e / c; const int ch = (int)(e - p * c);
        float acc = g1[e];
        for (int s = 0; s < ns; s++) {
            const long long r = p * ns + s;
            const float g = go[r * c + ch];
            acc += g;
            atomic_add_f32(g2 + (long long)idx[r] * c + ch, -g);
        }
        g1[e] = acc;
    }
}

// ---------------------------------------------------------------- K9 aggregation forward
// out[p,ch] += sum_s (in[idx[p,s],ch] + pos[p,s,ch]) * w[p,s,ch % w_c]    aggregation_cuda_kernel.cu:5-20
template <int V>
__global__ __launch_bounds__(GB) void agg_fwd(int n, int ns, int cv, int wcv, const float* __restrict__ in,
                                              const float* __restrict__ pos, const float* __restrict__ w,
                                              const int* __restrict__ idx, float* __restrict__ out)
{
    const long long total = (long long)n * cv;
    for (long long e = (long long)blockIdx.x * GB + threadIdx.x; e < total; e += (long long)gridDim.x * GB) {
        const long long p = e / cv; const int ch = (int)(e - p * cv);
        const int wch = ch % wcv;
        if (V == 4) {
            float4 acc = reinterpret_cast<float4*>(out)[e];
            for (int s = 0; s < ns; s++) {
                const long long r = p * ns + s;
                const float4 x = reinterpret_cast<const float4*>(in)[(long long)idx[r] * cv + ch];
                const float4 q = reinterpret_cast<const float4*>(pos)[r * cv + ch];
                const float4 ww = reinterpret_cast<const float4*>(w)[r * wcv + wch];
                acc.x += (x.x + q.x) * ww.x; acc.y += (x.y + q.y) * ww.y;
                acc.z += (x.z + q.z) * ww.z; acc.w += (x.w + q.w) * ww.w;
            }
            reinterpret_cast<float4*>(out)[e] = acc;
        } else {
            float acc = out[e];
            for (int s = 0; s < ns; s++) {
                const long long r = p * ns + s;
                acc += (in[(long long)idx[r] * cv + ch] + pos[r * cv + ch]) * w[r * wcv + wch];
            }
            out[e] = acc;
        }
    }
}

// with a processing order over the points (cbl_common.h)
__global__ __launch_bounds__(GB) void agg_fwd_v4_ordered(unsigned n, int ns, int cv, int wcv, const float4* __restrict__ in, const float4* __restrict__ pos,
                                                         const float4* __restrict__ w, const int* __restrict__ idx, const int* __restrict__ order,
                                                         float4* __restrict__ out)
{
    const unsigned long long total = (unsigned long long)n * cv;
    const unsigned nch = (unsigned)((total + GB - 1) / GB);
    for (unsigned v = blockIdx.x; v < 8 * cbl_xcd_per(nch); v += gridDim.x) {
        const unsigned long long e0 = (unsigned long long)cbl_xcd_slot(v, nch) * GB + threadIdx.x;
        if (e0 >= total) continue;
        const unsigned t = (unsigned)(e0 / (unsigned)cv), ch = (unsigned)(e0 - (unsigned long long)t * cv);
        const size_t p = (size_t)order[t], e = p * cv + ch;
        const unsigned wch = ch % (unsigned)wcv;
        float4 acc = out[e];
        for (int s = 0; s < ns; s++) {
            const size_t r = p * ns + s;
            const float4 x = in[(size_t)idx[r] * cv + ch];
            const float4 q = pos[r * cv + ch];
            const float4 ww = w[r * wcv + wch];
            acc.x += (x.x + q.x) * ww.x; acc.y += (x.y + q.y) * ww.y;
            acc.z += (x.z + q.z) * ww.z; acc.w += (x.w + q.w) * ww.w;
        }
        out[e] = acc;
    }
}

// ---------------------------------------------------------------- K10 aggregation backward
// gi[idx[p,s],ch] += g*w ; gpos[p,s,ch] = g*w ; gw[p,s,ch%w_c] += g*(in+pos)   aggregation_cuda_kernel.cu:22-39
__global__ __launch_bounds__(GB) void agg_bwd(int n, int ns, int c, int wc, const float* __restrict__ in,
                                              const float* __restrict__ pos, const float* __restrict__ w,
                                              const int* __restrict__ idx, const float* __restrict__ go,
                                              float* __restrict__ gi, float* __restrict__ gpos, float* __restrict__ gw)
{
    const long long total = (long long)n * c;
    for (long long e = (long long)blockIdx.x * GB + threadIdx.x; e < total; e += (long long)gridDim.x * GB) {
        const long long p = e / c; const int ch = (int)(e - p * c);
        const int wch = ch % wc;
        const float g = go[e];
        for (int s = 0; s < ns; s++) {
            const long long r = p * ns + s;
            const long long src = (long long)idx[r] * c + ch;
            const float wv = w[r * wc + wch];
            const float gwv = g * wv;
            if (gi) atomic_add_f32(gi + src, gwv);                   // gi == nullptr: the caller gathers it over the transposed table instead
            gpos[r * c + ch] = gwv;
            atomic_add_f32(gw + r * wc + wch, g * (in[src] + pos[r * c + ch]));
        }
    }
}

// Same, for the usual shapes (w_c a power of two dividing 64, c a multiple of 64 or a divisor of it): the c / w_c lanes of a point that
// share a weight column sit at stride w_c inside ONE wave, so their contributions to gw[p,s,ch % w_c] are summed across lanes
// first (log2 steps) and ONE lane adds them — the reference's one-atomic-per-(p,s,ch) scheme puts c / w_c = 8 atomics on every address.
__global__ __launch_bounds__(GB) void agg_bwd_wave(int n, int ns, int c, int wc, const float* __restrict__ in,
                                                   const float* __restrict__ pos, const float* __restrict__ w,
                                                   const int* __restrict__ idx, const float* __restrict__ go,
                                                   float* __restrict__ gi, float* __restrict__ gpos, float* __restrict__ gw)
{
    const long long total = (long long)n * c;
    const int span = c < 64 ? c : 64;                                // lanes of this wave that belong to the same point
    const long long e0 = (long long)blockIdx.x * GB + threadIdx.x;
    for (long long eb = e0 - (threadIdx.x & 63); eb < total; eb += (long long)gridDim.x * GB) {       // wave-uniform trip count
        const long long e = eb + (threadIdx.x & 63);
        const bool live = e < total;
        const long long p = live ? e / c : 0; const int ch = live ? (int)(e - p * c) : 0;
        const int wch = ch % wc;
        const float g = live ? go[e] : 0.f;
        for (int s = 0; s < ns; s++) {
            const long long r = p * ns + s;
            float contrib = 0.f;
            if (live) {
                const long long src = (long long)idx[r] * c + ch;
                const float gwv = g * w[r * wc + wch];
                if (gi) atomic_add_f32(gi + src, gwv);
                gpos[r * c + ch] = gwv;
                contrib = g * (in[src] + pos[r * c + ch]);
            }
            for (int st = wc; st < span; st <<= 1) contrib += __shfl_xor(contrib, st);
            if (live && (ch % span) < wc) atomic_add_f32(gw + r * wc + wch, contrib);
        }
    }
}

// ---------------------------------------------------------------- F1 queryandgroup (idx given)
// out[r, 0:3] = xyz[idx[r]] - new_xyz[r / ns] ; out[r, 3:3+c] = feat[idx[r]]     pointops.py:90-98
// A lane owns 4 channels of one output row: 16 B gather, 16 B store.  Output rows are 4*(3+c) B long, so the store is only
// 4-byte aligned: gfx950 global memory takes unaligned dwordx4 (the 4-byte-aligned vector type makes hipcc emit it); one extra lane per
// row writes the centred coordinates.  Scalar kernel for channel counts that are not a multiple of 4.
typedef float v4u __attribute__((ext_vector_type(4), aligned(4)));      // 16 B vector that may sit on any 4-byte boundary

__global__ __launch_bounds__(256) void query_group_v4(unsigned rows, int ns, int c4, int use_xyz,
                                                     const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                     const float4* __restrict__ feat, const int* __restrict__ idx, float* __restrict__ out)
{
    const unsigned parts = (unsigned)c4 + (use_xyz ? 1u : 0u);
    const unsigned oc = 4u * c4 + (use_xyz ? 3u : 0u);
    const unsigned long long total = (unsigned long long)rows * parts;
    for (unsigned long long e = (unsigned long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (unsigned long long)gridDim.x * 256) {
        const unsigned r = (unsigned)(e / parts), part = (unsigned)(e - (unsigned long long)r * parts);
        const unsigned src = (unsigned)idx[r];
        float* orow = out + (size_t)r * oc;
        if (part < (unsigned)c4) {
            const float4 v = feat[(size_t)src * c4 + part];
            __builtin_nontemporal_store(v4u{v.x, v.y, v.z, v.w}, reinterpret_cast<v4u*>(orow + (use_xyz ? 3 : 0) + 4 * part));   // streaming output
        } else {
            const unsigned q = r / (unsigned)ns;
            orow[0] = xyz[(size_t)src * 3 + 0] - new_xyz[(size_t)q * 3 + 0];
            orow[1] = xyz[(size_t)src * 3 + 1] - new_xyz[(size_t)q * 3 + 1];
            orow[2] = xyz[(size_t)src * 3 + 2] - new_xyz[(size_t)q * 3 + 2];
        }
    }
}

// use_xyz = 1: rows are 4*(3 + c) bytes long, so v4's 16-byte stores sit on 4-byte boundaries in three rows out of four.  Here a wave
// assembles a PIECE of consecutive output rows (16-byte aligned, a multiple of 16 bytes long) in LDS — aligned 16-byte gathers of the
// feature rows in, the centred coordinates from 3 lanes per row — and streams the piece out as ALIGNED, fully coalesced 16-byte stores.
// No 64-bit division per element (row / part come from the lane id).
//   order == nullptr: piece p = rows [16p, 16p + 16)
//   order != nullptr: piece t = the ns rows of query point order[t]; `order` lists the points in a spatially coherent sequence (the
//     cell order of the neighbour search, cbl_knnquery_ordered) and the pieces are dealt to the XCDs in CONTIGUOUS eighths (workgroup b
//     runs on XCD b % 8): the rows one XCD gathers then come from one slab of the scene and stay in its 4 MB L2.  With the scene's
//     own (shuffled) order every XCD reads the whole 10.5 MB feature table through the fabric: 157 MB fetched for a 190 MB kernel.
constexpr int QG_ROWS = 16, QG_MAX_ROWS = 32;
template <int C4T>                                                  // C4T > 0: channel count / 4 known at compile time (gathers unrolled, all in flight); 0: any
__global__ __launch_bounds__(256) void query_group_lds(unsigned rows, int ns, int c4_rt, const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                       const float4* __restrict__ feat, const int* __restrict__ idx, const int* __restrict__ order,
                                                       float* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float qg_lds[];  // [4 waves][pr * oc]
    const int c4 = C4T ? C4T : c4_rt;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int oc = 4 * c4 + 3;
    const int pr = order ? ns : QG_ROWS;                            // rows per piece
    float* piece = qg_lds + (size_t)wv * pr * oc;
    const unsigned npieces = rows / (unsigned)pr;                   // full pieces; remaining rows (unordered mode only) by the lane loop below
    const int nf4 = pr * c4;                                        // 16-byte feature parts per piece
    const int nchunk = pr * oc / 4;                                 // 16-byte chunks of the output piece (pr * oc is a multiple of 4)
    const unsigned nwg = (npieces + 3) >> 2;
    for (unsigned wg = blockIdx.x; wg < 8 * cbl_xcd_per(nwg); wg += gridDim.x) {
        const unsigned t = (order ? cbl_xcd_slot(wg, nwg) : wg) * 4 + wv;       // XCD-contiguous dealing of the sequence (cbl_common.h)
        if (t >= npieces) continue;                                 // wave-uniform
        const unsigned r0 = (order ? (unsigned)order[t] : t) * (unsigned)pr;
        const int myidx = idx[r0 + (lane < pr ? lane : 0)];           // lane l holds the support of row l
        // features: part f of the piece = (row f / c4, part f % c4); consecutive lanes read consecutive 16 B of a support row
        if constexpr (C4T > 0 && (QG_ROWS * C4T) % 64 == 0) {
            constexpr int NL = QG_MAX_ROWS * C4T / 64;
            float4 v[NL];
#pragma unroll
            for (int j = 0; j < NL; j++) {
                const int f = lane + 64 * j, row = f / C4T, part = f % C4T;
                const int src = __shfl(myidx, row < pr ? row : 0);  // the shuffle stays outside the lane-dependent branch: every source lane is active
                if (f < nf4) v[j] = feat[(size_t)src * C4T + part];
            }
#pragma unroll
            for (int j = 0; j < NL; j++) {
                const int f = lane + 64 * j, row = f / C4T, part = f % C4T;
                if (f < nf4) {
                    float* d = piece + row * oc + 3 + 4 * part;
                    d[0] = v[j].x; d[1] = v[j].y; d[2] = v[j].z; d[3] = v[j].w;
                }
            }
        } else {
            for (int f0 = 0; f0 < nf4; f0 += 64) {                    // wave-uniform trip count
                const int f = f0 + lane, row = f / c4, part = f - row * c4;
                const int src = __shfl(myidx, row < pr ? row : 0);
                if (f < nf4) {
                    const float4 v = feat[(size_t)src * c4 + part];
                    float* d = piece + row * oc + 3 + 4 * part;
                    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                }
            }
        }
        for (int e0 = 0; e0 < 3 * pr; e0 += 64) {                   // centred coordinates: e = 3 * row + axis
            const int e = e0 + lane, row = e / 3, a = e - 3 * row;
            const int src = __shfl(myidx, row < pr ? row : 0);
            if (e < 3 * pr) {
                const unsigned q = (r0 + row) / (unsigned)ns;
                piece[row * oc + a] = xyz[(size_t)src * 3 + a] - new_xyz[(size_t)q * 3 + a];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        typedef float v4a __attribute__((ext_vector_type(4)));          // 16-byte aligned vector (the builtin wants a plain vector type)
        v4a* o4 = reinterpret_cast<v4a*>(out + (size_t)r0 * oc);        // piece start: a multiple of pr * oc * 4 bytes, 16-byte aligned
        const v4a* p4 = reinterpret_cast<const v4a*>(piece);
        for (int ch = lane; ch < nchunk; ch += 64) __builtin_nontemporal_store(p4[ch], o4 + ch);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                            // the piece is reused by the next trip
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // tail rows (rows % 16, unordered mode), element by element
    const unsigned long long t0 = (unsigned long long)npieces * pr * oc, tot = (unsigned long long)rows * oc;
    for (unsigned long long e = t0 + (unsigned long long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (unsigned long long)gridDim.x * 256) {
        const unsigned r = (unsigned)(e / oc); const int ch = (int)(e - (unsigned long long)r * oc);
        const int src = idx[r];
        out[e] = ch < 3 ? xyz[(size_t)src * 3 + ch] - new_xyz[(size_t)(r / (unsigned)ns) * 3 + ch]
                        : reinterpret_cast<const float*>(feat)[(size_t)src * 4 * c4 + (ch - 3)];
    }
}

#include "query_group_pipe.h"

__global__ __launch_bounds__(GB) void query_group(long long rows, int ns, int c, int use_xyz,
                                                  const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                  const float* __restrict__ feat, const int* __restrict__ idx, float* __restrict__ out)
{
    const int oc = c + (use_xyz ? 3 : 0);
    const long long total = rows * oc;
    for (long long e = (long long)blockIdx.x * GB + threadIdx.x; e < total; e += (long long)gridDim.x * GB) {
        const long long r = e / oc; int ch = (int)(e - r * oc);
        const long long src = idx[r];
        float v;
        if (use_xyz) {
            if (ch < 3) v = xyz[src * 3 + ch] - new_xyz[(r / ns) * 3 + ch];
            else        v = feat[src * c + (ch - 3)];
        } else v = feat[src * c + ch];
        out[e] = v;
    }
}

// ---------------------------------------------------------------- F4 interpolation weights
// dist = sqrt(dist2); r = 1/(dist + 1e-8); w = r / sum_i r                       pointops.py:170-173
__global__ __launch_bounds__(GB) void interp_weights(int n, int k, const float* __restrict__ dist2,
                                                     float* __restrict__ weight, float* __restrict__ dist)
{
    for (long long p = (long long)blockIdx.x * GB + threadIdx.x; p < n; p += (long long)gridDim.x * GB) {
        float norm = 0.f;
        for (int i = 0; i < k; i++) {
            const float d = sqrtf(dist2[p * k + i]);          // correctly rounded (hipcc default)
            if (dist) dist[p * k + i] = d;
            const float r = 1.0f / (d + 1e-8f);
            weight[p * k + i] = r;
            norm += r;                                                // torch.sum over k, ascending
        }
        for (int i = 0; i < k; i++) weight[p * k + i] = weight[p * k + i] / norm;
    }
}

inline bool vec4_ok(int c, const void* a, const void* b, const void* d = nullptr, const void* e = nullptr)
{
    return (c % 4 == 0) && cbl_host_aligned16(a) && cbl_host_aligned16(b) && cbl_host_aligned16(d) && cbl_host_aligned16(e);
}

}  // namespace

#define CBL_CHECK_DIMS(...)   do { const long long _d[] = {__VA_ARGS__}; for (long long v : _d) if (v < 0) return CBL_ERR_BAD_ARG; } while (0)
#define CBL_CHECK_PTRS(...)   do { const void* _p[] = {__VA_ARGS__}; for (const void* v : _p) if (!v) return CBL_ERR_BAD_ARG; } while (0)

static int grouping_forward_impl(int m, int nsample, int c, const float* input, const int* idx, const int* order, float* output, void* stream)
{
    CBL_CHECK_DIMS(m, nsample, c);
    const long long rows = (long long)m * nsample;
    if (rows * c == 0) return CBL_OK;
    CBL_CHECK_PTRS(input, idx, output);
    hipStream_t st = cbl_stream(stream);
    if (order && vec4_ok(c, input, output) && rows * (c / 4) < 0xffffffffLL * GB)
        hipLaunchKernelGGL(grouping_fwd_v4_ordered, dim3(cbl_round_up8(cbl_grid_for(rows * (c / 4), GB, 256 * 64))), dim3(GB), 0, st, (unsigned)m, nsample, c / 4,
                           reinterpret_cast<const float4*>(input), idx, order, reinterpret_cast<float4*>(output));
    else if (vec4_ok(c, input, output))
        hipLaunchKernelGGL(grouping_fwd_v4, dim3(cbl_grid_for(rows * (c / 4), GB)), dim3(GB), 0, st, rows, c / 4,
                           reinterpret_cast<const float4*>(input), idx, reinterpret_cast<float4*>(output));
    else
        hipLaunchKernelGGL(grouping_fwd_s, dim3(cbl_grid_for(rows * c, GB)), dim3(GB), 0, st, rows, c, input, idx, output);
    return cbl_status();
}

CBL_EXPORT int cbl_grouping_forward(int m, int nsample, int c, const float* input, const int* idx, float* output, void* stream)
{
    return grouping_forward_impl(m, nsample, c, input, idx, nullptr, output, stream);
}

CBL_EXPORT int cbl_grouping_forward_ordered(int m, int nsample, int c, const float* input, const int* idx, const int* order, float* output, void* stream)
{
    return grouping_forward_impl(m, nsample, c, input, idx, order, output, stream);
}

CBL_EXPORT int cbl_grouping_backward(int m, int nsample, int c, const float* grad_output, const int* idx, float* grad_input, void* stream)
{
    CBL_CHECK_DIMS(m, nsample, c);
    const long long rows = (long long)m * nsample;
    if (rows * c == 0) return CBL_OK;
    CBL_CHECK_PTRS(grad_output, idx, grad_input);
    hipLaunchKernelGGL(grouping_bwd, dim3(cbl_grid_for(rows * c, GB)), dim3(GB), 0, cbl_stream(stream), rows, c, grad_output, idx, grad_input);
    return cbl_status();
}

CBL_EXPORT int cbl_interpolation_forward(int n, int c, int k, const float* input, const int* idx, const float* weight, float* output, void* stream)
{
    CBL_CHECK_DIMS(n, c, k);
    if ((long long)n * c == 0) return CBL_OK;
    CBL_CHECK_PTRS(input, idx, weight, output);
    hipStream_t st = cbl_stream(stream);
    if (vec4_ok(c, input, output))
        hipLaunchKernelGGL(interp_fwd<4>, dim3(cbl_grid_for((long long)n * (c / 4), GB)), dim3(GB), 0, st, n, c / 4, k, input, idx, weight, output);
    else
        hipLaunchKernelGGL(interp_fwd<1>, dim3(cbl_grid_for((long long)n * c, GB)), dim3(GB), 0, st, n, c, k, input, idx, weight, output);
    return cbl_status();
}

CBL_EXPORT int cbl_interpolation_backward(int n, int c, int k, const float* grad_output, const int* idx, const float* weight, float* grad_input, void* stream)
{
    CBL_CHECK_DIMS(n, c, k);
    if ((long long)n * c == 0) return CBL_OK;
    CBL_CHECK_PTRS(grad_output, idx, weight, grad_input);
    hipLaunchKernelGGL(interp_bwd, dim3(cbl_grid_for((long long)n * c, GB)), dim3(GB), 0, cbl_stream(stream), n, c, k, grad_output, idx, weight, grad_input);
    return cbl_status();
}

static int subtraction_forward_impl(int n, int nsample, int c, const float* input1, const float* input2, const int* idx, const int* order, float* output, void* stream)
{
    CBL_CHECK_DIMS(n, nsample, c);
    const long long rows = (long long)n * nsample;
    if (rows * c == 0) return CBL_OK;
    CBL_CHECK_PTRS(input1, input2, idx, output);
    hipStream_t st = cbl_stream(stream);
    if (order && vec4_ok(c, input1, input2, output) && rows * (c / 4) < 0xffffffffLL * GB)
        hipLaunchKernelGGL(sub_fwd_v4_ordered, dim3(cbl_round_up8(cbl_grid_for(rows * (c / 4), GB, 256 * 64))), dim3(GB), 0, st, (unsigned)n, nsample, c / 4,
                           reinterpret_cast<const float4*>(input1), reinterpret_cast<const float4*>(input2), idx, order, reinterpret_cast<float4*>(output));
    else if (vec4_ok(c, input1, input2, output))
        hipLaunchKernelGGL(sub_fwd<4>, dim3(cbl_grid_for(rows * (c / 4), GB)), dim3(GB), 0, st, rows, nsample, c / 4, input1, input2, idx, output);
    else
        hipLaunchKernelGGL(sub_fwd<1>, dim3(cbl_grid_for(rows * c, GB)), dim3(GB), 0, st, rows, nsample, c, input1, input2, idx, output);
    return cbl_status();
}

CBL_EXPORT int cbl_subtraction_forward(int n, int nsample, int c, const float* input1, const float* input2, const int* idx, float* output, void* stream)
{
    return subtraction_forward_impl(n, nsample, c, input1, input2, idx, nullptr, output, stream);
}

CBL_EXPORT int cbl_subtraction_forward_ordered(int n, int nsample, int c, const float* input1, const float* input2, const int* idx, const int* order,
                                               float* output, void* stream)
{
    return subtraction_forward_impl(n, nsample, c, input1, input2, idx, order, output, stream);
}

CBL_EXPORT int cbl_subtraction_backward(int n, int nsample, int c, const int* idx, const float* grad_output, float* grad_input1, float* grad_input2, void* stream)
{
    CBL_CHECK_DIMS(n, nsample, c);
    if ((long long)n * nsample * c == 0) return CBL_OK;
    CBL_CHECK_PTRS(idx, grad_output, grad_input1, grad_input2);
    hipLaunchKernelGGL(sub_bwd, dim3(cbl_grid_for((long long)n * c, GB)), dim3(GB), 0, cbl_stream(stream), n, nsample, c, idx, grad_output, grad_input1, grad_input2);
    return cbl_status();
}

static int aggregation_forward_impl(int n, int nsample, int c, int w_c, const float* input, const float* position, const float* weight, const int* idx,
                                    const int* order, float* output, void* stream)
{
    CBL_CHECK_DIMS(n, nsample, c, w_c);
    if ((long long)n * c == 0) return CBL_OK;
    if (w_c == 0) return CBL_ERR_BAD_ARG;
    CBL_CHECK_PTRS(input, position, weight, idx, output);
    hipStream_t st = cbl_stream(stream);
    if (order && vec4_ok(c, input, position, weight, output) && w_c % 4 == 0)
        hipLaunchKernelGGL(agg_fwd_v4_ordered, dim3(cbl_round_up8(cbl_grid_for((long long)n * (c / 4), GB, 256 * 64))), dim3(GB), 0, st, (unsigned)n, nsample, c / 4, w_c / 4,
                           reinterpret_cast<const float4*>(input), reinterpret_cast<const float4*>(position), reinterpret_cast<const float4*>(weight), idx, order,
                           reinterpret_cast<float4*>(output));
    else if (vec4_ok(c, input, position, weight, output) && w_c % 4 == 0)
        hipLaunchKernelGGL(agg_fwd<4>, dim3(cbl_grid_for((long long)n * (c / 4), GB)), dim3(GB), 0, st, n, nsample, c / 4, w_c / 4, input, position, weight, idx, output);
    else
        hipLaunchKernelGGL(agg_fwd<1>, dim3(cbl_grid_for((long long)n * c, GB)), dim3(GB), 0, st, n, nsample, c, w_c, input, position, weight, idx, output);
    return cbl_status();
}

CBL_EXPORT int cbl_aggregation_forward(int n, int nsample, int c, int w_c, const float* input, const float* position, const float* weight, const int* idx, float* output, void* stream)
{
    return aggregation_forward_impl(n, nsample, c, w_c, input, position, weight, idx, nullptr, output, stream);
}

CBL_EXPORT int cbl_aggregation_forward_ordered(int n, int nsample, int c, int w_c, const float* input, const float* position, const float* weight, const int* idx,
                                               const int* order, float* output, void* stream)
{
    return aggregation_forward_impl(n, nsample, c, w_c, input, position, weight, idx, order, output, stream);
}

CBL_EXPORT int cbl_aggregation_backward(int n, int nsample, int c, int w_c, const float* input, const float* position, const float* weight, const int* idx,
                                        const float* grad_output, float* grad_input, float* grad_position, float* grad_weight, void* stream)
{
    CBL_CHECK_DIMS(n, nsample, c, w_c);
    if ((long long)n * c == 0) return CBL_OK;
    if (w_c == 0) return CBL_ERR_BAD_ARG;
    CBL_CHECK_PTRS(input, position, weight, idx, grad_output, grad_position, grad_weight);   // grad_input may be NULL: cbl_weighted_scatter_csr writes it
    const bool pow2 = (w_c & (w_c - 1)) == 0;
    if (GB % 64 == 0 && pow2 && w_c <= 64 && c % w_c == 0 && (c % 64 == 0 || 64 % c == 0))
        hipLaunchKernelGGL(agg_bwd_wave, dim3(cbl_grid_for((long long)n * c, GB)), dim3(GB), 0, cbl_stream(stream), n, nsample, c, w_c, input, position, weight, idx, grad_output, grad_input, grad_position, grad_weight);
    else
        hipLaunchKernelGGL(agg_bwd, dim3(cbl_grid_for((long long)n * c, GB)), dim3(GB), 0, cbl_stream(stream), n, nsample, c, w_c, input, position, weight, idx, grad_output, grad_input, grad_position, grad_weight);
    return cbl_status();
}

static int queryandgroup_impl(int m, int nsample, int c, int use_xyz, const float* xyz, const float* new_xyz, const float* feat, const int* idx,
                              const int* order, float* out, void* stream)
{
    CBL_CHECK_DIMS(m, nsample, c);
    const long long rows = (long long)m * nsample;
    const int oc = c + (use_xyz ? 3 : 0);
    if (rows * oc == 0) return CBL_OK;
    CBL_CHECK_PTRS(idx, out);
    if (use_xyz) CBL_CHECK_PTRS(xyz, new_xyz);
    if (c > 0) CBL_CHECK_PTRS(feat);
    const bool lds_ok = use_xyz && c % 4 == 0 && c > 0 && c <= 128 && cbl_host_aligned16(feat) && cbl_host_aligned16(out) && rows < 0xffffffffLL;
    // the ordered form needs whole points as 16-byte aligned pieces: (nsample * (3 + c)) % 4 == 0, at most QG_MAX_ROWS rows
    if (order && !(lds_ok && nsample <= QG_MAX_ROWS && ((long long)nsample * oc) % 4 == 0 && sizeof(float) * 4 * (size_t)nsample * oc <= 65536)) order = nullptr;
    if (lds_ok && order && (c == 32 || c == 64) && (nsample == 8 || nsample == 16)) {
        // the networks' full-resolution shapes: persistent waves (as many workgroups as are resident at once), pieces = whole points
        const unsigned npieces = (unsigned)m, nwg = (npieces + 3) / 4;
        auto grid_of = [&](const void* fn, int) -> unsigned {           // workgroups resident at once (cbl_resident_blocks: cached per kernel and device, thread-safe)
            const unsigned g = cbl_resident_blocks(fn, GB, 0);
            return g < cbl_round_up8(nwg) ? g : cbl_round_up8(nwg);
        };
#define CBL_QG_PIPE(C4T, PR, SLOT) hipLaunchKernelGGL((query_group_lds_pipe<C4T, PR>), dim3(grid_of(reinterpret_cast<const void*>(&query_group_lds_pipe<C4T, PR>), SLOT)), dim3(GB), 0, \
                                                      cbl_stream(stream), npieces, xyz, new_xyz, reinterpret_cast<const float4*>(feat), idx, order, out)
        if (c == 32 && nsample == 8) CBL_QG_PIPE(8, 8, 0); else if (c == 32) CBL_QG_PIPE(8, 16, 1); else if (nsample == 8) CBL_QG_PIPE(16, 8, 2); else CBL_QG_PIPE(16, 16, 3);
#undef CBL_QG_PIPE
    } else if (lds_ok && (order || rows >= QG_ROWS)) {
        const int pr = order ? nsample : QG_ROWS;
        const dim3 grid(cbl_round_up8(cbl_grid_for((rows / pr + 3) / 4 * 256, GB, 256 * 64)));
        const size_t lds = sizeof(float) * 4 * (size_t)pr * (c + 3);
#define CBL_QG_LDS(C4T) hipLaunchKernelGGL(query_group_lds<C4T>, grid, dim3(GB), lds, cbl_stream(stream), (unsigned)rows, nsample, c / 4, xyz, new_xyz, \
                                           reinterpret_cast<const float4*>(feat), idx, order, out)
        switch (c) {
            case 32:  CBL_QG_LDS(8); break;
            case 64:  CBL_QG_LDS(16); break;
            case 128: CBL_QG_LDS(32); break;
            default:  CBL_QG_LDS(0); break;
        }
#undef CBL_QG_LDS
    } else if (c % 4 == 0 && c > 0 && cbl_host_aligned16(feat) && rows < 0xffffffffLL)
        hipLaunchKernelGGL(query_group_v4, dim3(cbl_grid_for(rows * (c / 4 + (use_xyz ? 1 : 0)), GB)), dim3(GB), 0, cbl_stream(stream), (unsigned)rows, nsample, c / 4, use_xyz,
                           xyz, new_xyz, reinterpret_cast<const float4*>(feat), idx, out);
    else
        hipLaunchKernelGGL(query_group, dim3(cbl_grid_for(rows * oc, GB)), dim3(GB), 0, cbl_stream(stream), rows, nsample, c, use_xyz, xyz, new_xyz, feat, idx, out);
    return cbl_status();
}

CBL_EXPORT int cbl_queryandgroup(int m, int nsample, int c, int use_xyz, const float* xyz, const float* new_xyz, const float* feat, const int* idx, float* out, void* stream)
{
    return queryandgroup_impl(m, nsample, c, use_xyz, xyz, new_xyz, feat, idx, nullptr, out, stream);
}

CBL_EXPORT int cbl_queryandgroup_ordered(int m, int nsample, int c, int use_xyz, const float* xyz, const float* new_xyz, const float* feat, const int* idx,
                                         const int* order, float* out, void* stream)
{
    return queryandgroup_impl(m, nsample, c, use_xyz, xyz, new_xyz, feat, idx, order, out, stream);
}

CBL_EXPORT int cbl_interpolation_weights(int n, int k, const float* dist2, float* weight, float* dist, void* stream)
{
    CBL_CHECK_DIMS(n, k);
    if ((long long)n * k == 0) return CBL_OK;
    CBL_CHECK_PTRS(dist2, weight);
    hipLaunchKernelGGL(interp_weights, dim3(cbl_grid_for(n, GB)), dim3(GB), 0, cbl_stream(stream), n, k, dist2, weight, dist);
    return cbl_status();
}
