// a14 / a15: TF-side local aggregation over radius neighbourhoods with a shadow (padding) index.
// Replaces the TF1 op chains of
//   PseudoGrid (KPConv, depthwise)   /root/reference/tensorflow/models/local_aggregation_operators.py:620-746 (math :681-728)
//   AdaptiveWeight                   ...:316-500 (shipped config: dp -> 1 FC -> weights, mean reduction with the :466-470 quirk)
//   ind_max_pool / ind_closest_pool  /root/reference/tensorflow/models/basic_operators.py:155-192
// Index convention of the TF side: neighbors_indices (n, K) int32, value == n0 (number of supports) means "no neighbour"
// and selects a shadow row (zeros for features, 1e6 / 0 for points, column-min for max pooling).
//
// MI355X mapping.  KPConv forward is the one GEMM-shaped piece of the hot path: per query point the influence matrix
// w (KP x K) times the gathered neighbour features (K x C).  One wave per point feeds it to the matrix cores as
// v_mfma_f32_16x16x4_f32 tiles (M = 16 kernel points, N = 16 channels, k = 4 neighbours): the A operand (influence
// weights) is COMPUTED in registers by the lane that owns (kernel point, neighbour) and never exists in memory, the B
// operand is the gathered feature row (64 B contiguous per neighbour), the accumulator tile is contracted with the
// depthwise kernel weights in the epilogue.  Exact f32 (fma chain), no reduced precision.  Nothing of shape (n,K,.) or
// (n,KP,.) is materialised, unlike the reference's gather / tile / matmul chain.  Backward passes and AdaptiveWeight are
// lane-per-channel VALU kernels (their contraction depth is 3..16), gradients of the shared parameters are reduced
// in registers over a persistent wave's points, then across the workgroup in LDS, then one L2 atomic per element.
#include "cbl_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// ---------------------------------------------------------------------------------------------- KPConv forward (MFMA)
// influence: 0 constant, 1 linear.  closest: only the nearest kernel point of each neighbour keeps its weight (:705-708).
//
// Tile mapping (v_mfma_f32_16x16x4_f32: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15], D[row = (lane>>4)*4 + r][col = lane&15]):
//   i / row = kernel point, k = neighbour within a group of 4, and — to make the B operand ONE 16-byte load — column j of
//   accumulator tile t stands for channel c0 + 4*j + t.  Lane (k, j) then loads the float4 f[nbr_k][c0 + 4j .. 4j+3] (16 lanes
//   = one 256 B row segment per neighbour, 1 KiB per wave-instruction) and its four components are the B values of the four
//   tiles; after the epilogue lane j holds channels c0 + 4j .. 4j+3 of the output, stored as one float4 (256 B per point).
template <bool VEC>       // VEC: C % 4 == 0 and 16-byte aligned rows -> float4 path; else scalar loads with the same mapping
__global__ __launch_bounds__(256) void kpconv_fwd_kernel(int n, int n0, int K, int C, int KP, const float* __restrict__ q,
                                                         const float* __restrict__ s, const int* __restrict__ idx, const float* __restrict__ f,
                                                         const float* __restrict__ kpts, const float* __restrict__ kw, float extent,
                                                         int influence, int closest, const int* __restrict__ order, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int kp_id = lane & 15;          // A row / kernel point; also B / D column j
    const int kq = lane >> 4;             // neighbour within a group of 4; D row block
    const unsigned wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwg = ((unsigned)n + 3u) >> 2;
    const bool kp_ok = kp_id < KP;
    const float kx = kp_ok ? kpts[3 * kp_id] : 0.f, ky = kp_ok ? kpts[3 * kp_id + 1] : 0.f, kz = kp_ok ? kpts[3 * kp_id + 2] : 0.f;

    const float inv_extent = 1.0f / extent;
    // one point per wave and trip; with `order` the points are taken in that sequence, dealt to the XCDs in contiguous eighths (cbl_common.h).
    // The trip is software-pipelined over its three dependent round trips (processing slot -> point id -> neighbour ids -> neighbour
    // coordinates): while point i is gathered and multiplied, the ids of point i+1 and the point id of i+2 are in flight, and the
    // coordinates of i+1 are requested at the end of the trip — the kernel was bound by those round trips (5 points per wave in a row).
    const unsigned vend = 8 * cbl_xcd_per(nwg), vstep = gridDim.x;
    auto point_at = [&](unsigned v) -> int {
        if (v >= vend) return -1;
        const unsigned t = (order ? cbl_xcd_slot(v, nwg) : v) * 4 + wv;
        return t < (unsigned)n ? (order ? order[t] : (int)t) : -1;
    };
    struct Geo { int id; float sx, sy, sz, qx, qy, qz; };
    auto fetch_ids = [&](int pt, Geo& g) {                          // first round trip of a point: its query coordinates and neighbour ids
        g.id = (lane < K) ? idx[(size_t)pt * K + lane] : n0;
        g.qx = q[3 * pt]; g.qy = q[3 * pt + 1]; g.qz = q[3 * pt + 2];
    };
    auto fetch_xyz = [&](Geo& g) {                                  // second: the neighbours' coordinates; the shadow point sits at (1e6,1e6,1e6)  (:681-684)
        const bool real = g.id >= 0 && g.id < n0;
        // one 12-byte load per lane (three dword loads cost the texture path three passes over the 64 scattered addresses), unconditional
        // with a clamped row, masked afterwards
        const float3 sp = *reinterpret_cast<const float3*>(s + 3 * (size_t)(real ? g.id : 0));
        g.sx = real ? sp.x : 1e6f; g.sy = real ? sp.y : 1e6f; g.sz = real ? sp.z : 1e6f;
    };
    int p0 = point_at(blockIdx.x), p1 = point_at(blockIdx.x + vstep);
    Geo cur = {n0, 1e6f, 1e6f, 1e6f, 0.f, 0.f, 0.f};
    if (p0 >= 0) { fetch_ids(p0, cur); fetch_xyz(cur); }
    for (unsigned v = blockIdx.x; v < vend; v += vstep) {
        const int p2 = point_at(v + 2 * vstep);                     // point id two trips ahead
        Geo nxt = {n0, 1e6f, 1e6f, 1e6f, 0.f, 0.f, 0.f};
        if (p1 >= 0) fetch_ids(p1, nxt);                            // neighbour ids of the next point
        const int p = p0;
        const Geo g = cur;
        auto advance = [&]() { if (p1 >= 0) fetch_xyz(nxt); cur = nxt; p0 = p1; p1 = p2; };
        if (p < 0) { advance(); continue; }
        const float qx = g.qx, qy = g.qy, qz = g.qz;
        for (int c0 = 0; c0 < C; c0 += 64) {
            f32x4 acc[4];
#pragma unroll
            for (int t = 0; t < 4; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int cb = c0 + 4 * kp_id;                                                   // first of this lane's 4 channels
            for (int k0 = 0; k0 < K; k0 += 64) {
                // up to 64 neighbour ids and their coordinates per chunk: prefetched for the first chunk, loaded here for K > 64
                int my_id; float mrx, mry, mrz;
                if (k0 == 0) { my_id = g.id; mrx = g.sx - qx; mry = g.sy - qy; mrz = g.sz - qz; }
                else {
                    const int my_nb = k0 + lane;
                    my_id = (my_nb < K) ? idx[(size_t)p * K + my_nb] : n0;
                    const bool my_real = my_id >= 0 && my_id < n0;
                    mrx = (my_real ? s[3 * my_id] : 1e6f) - qx; mry = (my_real ? s[3 * my_id + 1] : 1e6f) - qy; mrz = (my_real ? s[3 * my_id + 2] : 1e6f) - qz;
                }
                const int kend = min(64, K - k0);
                for (int kc = 0; kc < kend; kc += 16) {
                    // feature rows of 4 groups of 4 neighbours are requested before any of them is consumed
                    float bv[4][4]; float a[4];
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const int src = kc + 4 * g + kq;                                         // lane holding this neighbour
                        const int id = __shfl(my_id, src & 63);
                        const bool in = (kc + 4 * g + kq) < kend;
                        const bool real = in && id >= 0 && id < n0;
#pragma unroll
                        for (int t = 0; t < 4; t++) bv[g][t] = 0.f;                              // shadow feature row = 0 (:713)
                        if (real) {
                            if (VEC) {
                                if (cb < C) { const float4 v = *reinterpret_cast<const float4*>(f + (size_t)id * C + cb); bv[g][0] = v.x; bv[g][1] = v.y; bv[g][2] = v.z; bv[g][3] = v.w; }
                            } else {
#pragma unroll
                                for (int t = 0; t < 4; t++) if (cb + t < C) bv[g][t] = f[(size_t)id * C + cb + t];
                            }
                        }
                        const float rx = __shfl(mrx, src & 63), ry = __shfl(mry, src & 63), rz = __shfl(mrz, src & 63);
                        const float dx = rx - kx, dy = ry - ky, dz = rz - kz;
                        const float sq = (dx * dx + dy * dy) + dz * dz;                          // :688
                        // v_sqrt_f32 (1 ulp) and a multiply by 1/extent instead of the correctly rounded sqrt and division (~25 VALU per weight):
                        // 1e-7 relative on w, the contract is 1e-4
                        float w = influence ? fmaxf(1.0f - __builtin_amdgcn_sqrtf(sq) * inv_extent, 0.0f) : 1.0f;    // :697 / :693
                        if (closest) {                                                           // argmin over kernel points, first minimum
                            float bs = kp_ok ? sq : INFINITY; int bi = kp_id;
#pragma unroll
                            for (int sft = 8; sft >= 1; sft >>= 1) {
                                const float os = __shfl_xor(bs, sft, 16); const int oi = __shfl_xor(bi, sft, 16);
                                if (os < bs || (os == bs && oi < bi)) { bs = os; bi = oi; }
                            }
                            if (bi != kp_id) w = 0.f;
                        }
                        a[g] = (kp_ok && in) ? w : 0.f;
                    }
#pragma unroll
                    for (int g = 0; g < 4; g++)
#pragma unroll
                        for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g], bv[g][t], acc[t], 0, 0, 0);   // wf = w @ f_nbr (:716)
                }
            }
            // epilogue: out[c] = sum_kp kernel_weights[kp,c] * wf[kp,c]  (:723-727); tile t, column j <-> channel cb + t
            float res[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                float part = 0.f;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = kq * 4 + r;
                    const float kwv = (row < KP && cb + t < C) ? kw[(size_t)row * C + cb + t] : 0.f;
                    part += kwv * acc[t][r];
                }
                part += __shfl_xor(part, 16);
                part += __shfl_xor(part, 32);
                res[t] = part;
            }
            if (lane < 16) {
                if (VEC) { if (cb < C) *reinterpret_cast<float4*>(out + (size_t)p * C + cb) = make_float4(res[0], res[1], res[2], res[3]); }
                else {
#pragma unroll
                    for (int t = 0; t < 4; t++) if (cb + t < C) out[(size_t)p * C + cb + t] = res[t];
                }
            }
        }
        advance();                                                  // coordinates of the next point's neighbours: their ids have arrived by now
    }
}

// The same kernel for the shapes the networks use (C <= 64, C % 4 == 0, 16-byte rows, K <= 64), without a branch in the trip: the feature rows
// are loaded unconditionally from a clamped row and masked afterwards (an exec-masked load per neighbour group made the compiler order every
// group's load behind the previous group's wait), the kernel weights of the epilogue (the same for every point) are read once into registers.
// sum over the four 16-lane rows of a wave, on the vector ALU (gfx950 v_permlane16_swap / v_permlane32_swap) instead of two ds_bpermute round trips
__device__ __forceinline__ float rows_sum4(float x)
{
    const unsigned u = __float_as_uint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);            // {rows 0,0,2,2} , {rows 1,1,3,3}
    const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const unsigned w = __float_as_uint(y);
    const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);            // {lower half twice} , {upper half twice}
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <bool CLOSEST, bool LINEAR, bool ONE>          // ONE: K <= 16, a single chunk of 16 neighbours: the first MFMAs start from a literal 0
__global__ __launch_bounds__(256) void kpconv_fwd_c64_kernel(int n, int n0, int K, int C, int KP, const float* __restrict__ q,
                                                             const float* __restrict__ s, const int* __restrict__ idx, const float* __restrict__ f,
                                                             const float* __restrict__ kpts, const float* __restrict__ kw, float extent,
                                                             const int* __restrict__ order, float* __restrict__ out)
{
    // epilogue operand kernel_weights (KP x C), the same for every point: once per workgroup into LDS (a ds_read is ~100 clocks behind the MFMAs, a
    // global read of the same L1-resident rows ~500; held in registers across the trip they cost two waves per SIMD)
    __shared__ float4 kw_s[16 * 16];
    {
        const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
        kw_s[threadIdx.x] = (row < KP && 4 * col < C) ? *reinterpret_cast<const float4*>(kw + (size_t)row * C + 4 * col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int kp_id = lane & 15, kq = lane >> 4;
    const unsigned wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwg = ((unsigned)n + 3u) >> 2;
    const bool kp_ok = kp_id < KP;
    const float kx = kp_ok ? kpts[3 * kp_id] : 0.f, ky = kp_ok ? kpts[3 * kp_id + 1] : 0.f, kz = kp_ok ? kpts[3 * kp_id + 2] : 0.f;
    const float inv_extent = 1.0f / extent;
    const float m2kx = -2.f * kx, m2ky = -2.f * ky, m2kz = -2.f * kz, k2 = (kx * kx + ky * ky) + kz * kz;
    const int cb = 4 * kp_id;
    const bool ch_ok = cb < C;
    const char* fbytes = reinterpret_cast<const char*>(f);
    const unsigned row_bytes = 4u * (unsigned)C, col_bytes = 4u * (unsigned)(ch_ok ? cb : 0);
    // Accumulator rows kp >= KP have weight 0 (A operand), channel columns >= C are never stored and never mix with others, the rows of neighbours
    // that are not real are zeroed behind the (clamped, unconditional) load: the loads only have to be readable.
    const unsigned vend = 8 * cbl_xcd_per(nwg), vstep = gridDim.x;
    auto point_at = [&](unsigned v) -> int {
        if (v >= vend) return -1;
        const unsigned t = (order ? cbl_xcd_slot(v, nwg) : v) * 4 + wv;
        return t < (unsigned)n ? (order ? order[t] : (int)t) : -1;
    };
    // Three points in flight per wave, every load one trip ahead of its use: while point i is multiplied, the feature rows and the neighbour
    // coordinates of point i+1 (its ids arrived during the previous trip) and the neighbour ids of point i+2 are on their way.  The kernel is bound by
    // these round trips, not by its instruction stream (PMC: halving the vector instructions moved it by 3 %).
    struct Xyz { float x, y, z, qx, qy, qz; };
    struct Rows { float4 v[4]; bool real[4]; };
    auto load_ids = [&](int pt) -> int { return (pt >= 0 && lane < K) ? idx[(size_t)pt * K + lane] : n0; };
    auto load_xyz = [&](int pt, int id) -> Xyz {                     // the shadow point sits at (1e6,1e6,1e6)  (:681-684)
        Xyz g;
        const bool real = id >= 0 && id < n0;
        const float3 sp = *reinterpret_cast<const float3*>(s + 3 * (size_t)(real ? id : 0));       // one 12-byte load, clamped row, masked afterwards
        g.x = real ? sp.x : 1e6f; g.y = real ? sp.y : 1e6f; g.z = real ? sp.z : 1e6f;
        const float3 qp = *reinterpret_cast<const float3*>(q + 3 * (size_t)(pt >= 0 ? pt : 0));
        g.qx = qp.x; g.qy = qp.y; g.qz = qp.z;
        return g;
    };
    auto load_rows = [&](int ids, int kc) -> Rows {                  // 16 neighbours' rows: lane (kq, j) takes channels 4j.. of neighbours kc + 4 gi + kq
        Rows r;
#pragma unroll
        for (int gi = 0; gi < 4; gi++) {
            const int id = __shfl(ids, (kc + 4 * gi + kq) & 63);     // lanes >= K hold the shadow id
            r.real[gi] = id >= 0 && id < n0;
            r.v[gi] = *reinterpret_cast<const float4*>(fbytes + ((unsigned)(r.real[gi] ? id : 0) * row_bytes + col_bytes));   // 32-bit offsets: n0 * C * 4 < 2^32 (host)
        }
        return r;
    };
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    auto chunk = [&](const Rows& rows, int kc, float mrx, float mry, float mrz, float mr2, f32x4 (&acc)[4], bool first) {
        float a[4];
#pragma unroll
        for (int gi = 0; gi < 4; gi++) {
            const int src = (kc + 4 * gi + kq) & 63;
            const float rx = __shfl(mrx, src), ry = __shfl(mry, src), rz = __shfl(mrz, src);
            float sq;
            if (CLOSEST) {                                           // the argmin compares sq exactly: the reference's association (:688)
                const float dx = rx - kx, dy = ry - ky, dz = rz - kz;
                sq = (dx * dx + dy * dy) + dz * dz;
            } else {
                // |r - k|^2 = |r|^2 + |k|^2 - 2 r.k as one add + three fma; r and k are offsets within one neighbourhood (a few extents), so the
                // cancellation costs ~1e-6 relative to extent^2: the contract is 1e-4
                const float r2 = __shfl(mr2, src);
                sq = fmaxf(__builtin_fmaf(m2kx, rx, __builtin_fmaf(m2ky, ry, __builtin_fmaf(m2kz, rz, r2 + k2))), 0.f);
            }
            float w = LINEAR ? fmaxf(1.0f - __builtin_amdgcn_sqrtf(sq) * inv_extent, 0.0f) : 1.0f;    // :697 / :693
            if (CLOSEST) {                                           // argmin over kernel points, first minimum
                float bs = kp_ok ? sq : INFINITY; int bi = kp_id;
#pragma unroll
                for (int sft = 8; sft >= 1; sft >>= 1) {
                    const float os = __shfl_xor(bs, sft, 16); const int oi = __shfl_xor(bi, sft, 16);
                    if (os < bs || (os == bs && oi < bi)) { bs = os; bi = oi; }
                }
                if (bi != kp_id) w = 0.f;
            }
            a[gi] = (kp_ok && rows.real[gi]) ? w : 0.f;              // shadow feature row = 0 (:713): its weight is
        }
#pragma unroll
        for (int gi = 0; gi < 4; gi++) {
            const bool z = first && gi == 0;
            // a neighbour that is not real reads the clamped row 0 of the table: its weight is already 0, the row is zeroed as well so that a
            // non-finite value there cannot leak (0 * inf) — four selects per group on a kernel that is not bound by its vector instructions
            const bool rl = rows.real[gi];
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[gi], rl ? rows.v[gi].x : 0.f, z ? zero : acc[0], 0, 0, 0);   // wf = w @ f_nbr (:716)
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[gi], rl ? rows.v[gi].y : 0.f, z ? zero : acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[gi], rl ? rows.v[gi].z : 0.f, z ? zero : acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[gi], rl ? rows.v[gi].w : 0.f, z ? zero : acc[3], 0, 0, 0);
        }
    };
    int p0 = point_at(blockIdx.x), p1 = point_at(blockIdx.x + vstep), p2 = point_at(blockIdx.x + 2 * vstep);
    int id0 = load_ids(p0), id1 = load_ids(p1);
    Xyz g0 = load_xyz(p0, id0);
    Rows r0 = load_rows(id0, 0);
    for (unsigned v = blockIdx.x; v < vend; v += vstep) {
        const int p3 = point_at(v + 3 * vstep);
        const int p = p0;
        // next point's rows and coordinates, the ids of the one after: all requested before anything of this point is waited for
        const int id2 = load_ids(p2); const Xyz g1 = load_xyz(p1, id1); const Rows r1 = load_rows(id1, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (p >= 0) {                                                // (an empty slot: the XCD dealing leaves holes before the end)
            const float mrx = g0.x - g0.qx, mry = g0.y - g0.qy, mrz = g0.z - g0.qz;
            const float mr2 = (mrx * mrx + mry * mry) + mrz * mrz;
            f32x4 acc[4];
#pragma unroll
            for (int t = 0; t < 4; t++) acc[t] = zero;
            chunk(r0, 0, mrx, mry, mrz, mr2, acc, true);
            if (!ONE) {
                for (int k0 = 0; k0 < K; k0 += 64) {                  // K > 16: the remaining chunks, loaded where they are used
                    int ids = id0; float cx = mrx, cy = mry, cz = mrz, c2 = mr2;
                    if (k0 > 0) {
                        ids = (k0 + lane < K) ? idx[(size_t)p * K + k0 + lane] : n0;
                        const bool real = ids >= 0 && ids < n0;
                        const float3 sp = *reinterpret_cast<const float3*>(s + 3 * (size_t)(real ? ids : 0));
                        cx = (real ? sp.x : 1e6f) - g0.qx; cy = (real ? sp.y : 1e6f) - g0.qy; cz = (real ? sp.z : 1e6f) - g0.qz;
                        c2 = (cx * cx + cy * cy) + cz * cz;
                    }
                    for (int kc = (k0 == 0 ? 16 : 0); kc < min(64, K - k0); kc += 16) {
                        const Rows rr = load_rows(ids, kc);
                        chunk(rr, kc, cx, cy, cz, c2, acc, false);
                    }
                }
            }
            float res[4];
#pragma unroll
            for (int t = 0; t < 4; t++) res[t] = 0.f;
#pragma unroll
            for (int r = 0; r < 4; r++) {                            // out[c] = sum_kp kernel_weights[kp,c] * wf[kp,c]  (:723-727); tile t, column j <-> channel cb + t
                const float4 k4 = kw_s[(kq * 4 + r) * 16 + kp_id];
                res[0] = __builtin_fmaf(k4.x, acc[0][r], res[0]); res[1] = __builtin_fmaf(k4.y, acc[1][r], res[1]);
                res[2] = __builtin_fmaf(k4.z, acc[2][r], res[2]); res[3] = __builtin_fmaf(k4.w, acc[3][r], res[3]);
            }
#pragma unroll
            for (int t = 0; t < 4; t++) res[t] = rows_sum4(res[t]);
            if (lane < 16 && ch_ok) *reinterpret_cast<float4*>(out + (size_t)p * C + cb) = make_float4(res[0], res[1], res[2], res[3]);
        }
        p0 = p1; p1 = p2; p2 = p3; id0 = id1; id1 = id2; g0 = g1; r0 = r1;
    }
}

// ---------------------------------------------------------------------------------------------- KPConv backward (VALU)
// grad_features[nbr_k, c] += go[c] * sum_kp w[kp,k] * kw[kp,c]      grad_kw[kp,c] += go[c] * sum_k w[kp,k] * f[nbr_k, c]
// one wave per point; lanes first build w (KP x K) in LDS, then lane = channel.
constexpr int KPB_MAXKP = 16, KPB_MAXK = 64;
__global__ __launch_bounds__(256) void kpconv_bwd_kernel(int n, int n0, int K, int C, int KP, const float* __restrict__ q,
                                                         const float* __restrict__ s, const int* __restrict__ idx, const float* __restrict__ f,
                                                         const float* __restrict__ kpts, const float* __restrict__ kw, float extent,
                                                         int influence, int closest, const float* __restrict__ go,
                                                         float* __restrict__ gf, float* __restrict__ gkw)
{
    __shared__ float w_s[4][KPB_MAXKP * KPB_MAXK];
    __shared__ int id_s[4][KPB_MAXK];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wave0 = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6), nwaves = (gridDim.x * 256) >> 6;
    float* W = w_s[wv]; int* ID = id_s[wv];
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        const bool cok = c < C;
        float kwr[KPB_MAXKP], gacc[KPB_MAXKP];
#pragma unroll
        for (int kp = 0; kp < KPB_MAXKP; kp++) { kwr[kp] = (cok && kp < KP) ? kw[(size_t)kp * C + c] : 0.f; gacc[kp] = 0.f; }
        for (int p = wave0; p < n; p += nwaves) {
            const float qx = q[3 * p], qy = q[3 * p + 1], qz = q[3 * p + 2];
            // phase 1: neighbour ids and the influence matrix
            for (int k = lane; k < K; k += 64) ID[k] = idx[(size_t)p * K + k];
            for (int e = lane; e < KP * K; e += 64) {
                const int kp = e / K, k = e - kp * K;
                const int id = idx[(size_t)p * K + k];
                const bool real = id >= 0 && id < n0;
                const float rx = (real ? s[3 * id] : 1e6f) - qx, ry = (real ? s[3 * id + 1] : 1e6f) - qy, rz = (real ? s[3 * id + 2] : 1e6f) - qz;
                const float dx = rx - kpts[3 * kp], dy = ry - kpts[3 * kp + 1], dz = rz - kpts[3 * kp + 2];
                const float sq = (dx * dx + dy * dy) + dz * dz;
                float w = influence ? fmaxf(1.0f - sqrtf(sq) / extent, 0.0f) : 1.0f;
                if (closest) {
                    for (int o = 0; o < KP; o++) {
                        const float ex = rx - kpts[3 * o], ey = ry - kpts[3 * o + 1], ez = rz - kpts[3 * o + 2];
                        const float osq = (ex * ex + ey * ey) + ez * ez;
                        if (osq < sq || (osq == sq && o < kp)) { w = 0.f; break; }
                    }
                }
                W[kp * K + k] = w;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): this wave's LDS writes have landed
            __builtin_amdgcn_wave_barrier();               // and the compiler keeps the reads below after them
            const float g = cok ? go[(size_t)p * C + c] : 0.f;
            float wf[KPB_MAXKP];
#pragma unroll
            for (int kp = 0; kp < KPB_MAXKP; kp++) wf[kp] = 0.f;
            for (int k = 0; k < K; k++) {
                const int id = ID[k];
                const bool real = id >= 0 && id < n0;
                const float fk = (real && cok) ? f[(size_t)id * C + c] : 0.f;
                float coef = 0.f;
#pragma unroll
                for (int kp = 0; kp < KPB_MAXKP; kp++) {
                    const float w = (kp < KP) ? W[kp * K + k] : 0.f;
                    wf[kp] += w * fk;
                    coef += w * kwr[kp];
                }
                if (real && cok && gf) unsafeAtomicAdd(gf + (size_t)id * C + c, g * coef);
            }
#pragma unroll
            for (int kp = 0; kp < KPB_MAXKP; kp++) gacc[kp] += g * wf[kp];
        }
        if (gkw && cok)
            for (int kp = 0; kp < KP; kp++) unsafeAtomicAdd(gkw + (size_t)kp * C + c, gacc[kp]);
    }
}

// ---------------------------------------------------------------------------------------------- AdaptiveWeight
// agg[p,c] = (1/nn[p]) * sum_k (rel[p,k,:] . fcw[:,c] + fcb[c]) * f[nbr_k, c],  rel = (s[nbr] - q[p]) / radius, shadow point = 0
// nn[p] = #{k : idx[p,k] < max(idx)} + 1e-5  ("mean" reduction, :466-470);  reduction_mean = 0 -> plain sum
// one atomic per workgroup (a same-address atomicMax per wave was the whole cost: 4096 of them serialised in L2, 46 us for 5 M indices), 16-byte loads
__global__ __launch_bounds__(256) void index_max_kernel(long long total, int vec, const int* __restrict__ idx, int* __restrict__ out)
{
    __shared__ int red[4];
    int m = -2147483647 - 1;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
    const long long t4 = vec ? (total >> 2) : 0;
    const int4* __restrict__ v = reinterpret_cast<const int4*>(idx);
    for (long long e = gid; e < t4; e += stride) { const int4 x = v[e]; m = max(m, max(max(x.x, x.y), max(x.z, x.w))); }
    for (long long e = 4 * t4 + gid; e < total; e += stride) m = max(m, idx[e]);
    for (int s = 32; s >= 1; s >>= 1) m = max(m, __shfl_xor(m, s));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, max(max(red[0], red[1]), max(red[2], red[3])));
}

template <bool BWD>
__global__ __launch_bounds__(256) void adaptive_weight_kernel(int n, int n0, int K, int C, const float* __restrict__ q, const float* __restrict__ s,
                                                              const int* __restrict__ idx, const float* __restrict__ f, float radius,
                                                              const float* __restrict__ fcw, const float* __restrict__ fcb,
                                                              const int* __restrict__ padding_num, int reduction_mean,
                                                              float* __restrict__ out,                       // forward
                                                              const float* __restrict__ go, float* __restrict__ gf,
                                                              float* __restrict__ gfcw, float* __restrict__ gfcb)   // backward
{
    const int lane = threadIdx.x & 63;
    const int wave0 = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6), nwaves = (gridDim.x * 256) >> 6;
    const int pad = reduction_mean ? *padding_num : 0;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        const bool cok = c < C;
        const float w0 = cok ? fcw[c] : 0.f, w1 = cok ? fcw[C + c] : 0.f, w2 = cok ? fcw[2 * C + c] : 0.f, bb = cok ? fcb[c] : 0.f;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, ab = 0.f;                      // parameter gradients of this channel
        for (int p = wave0; p < n; p += nwaves) {
            const float qx = q[3 * p], qy = q[3 * p + 1], qz = q[3 * p + 2];
            float nn = 1.f;
            if (reduction_mean) {
                int cnt = 0;
                for (int k = lane; k < K; k += 64) cnt += idx[(size_t)p * K + k] < pad ? 1 : 0;
                for (int sft = 32; sft >= 1; sft >>= 1) cnt += __shfl_xor(cnt, sft);
                nn = (float)cnt + 1e-5f;
            }
            const float g = (BWD && cok) ? go[(size_t)p * C + c] / nn : 0.f;
            float acc = 0.f;
            for (int k = 0; k < K; k++) {
                const int id = idx[(size_t)p * K + k];                      // wave-uniform
                const bool real = id >= 0 && id < n0;
                const float rx = ((real ? s[3 * id] : 0.f) - qx) / radius, ry = ((real ? s[3 * id + 1] : 0.f) - qy) / radius,
                            rz = ((real ? s[3 * id + 2] : 0.f) - qz) / radius;                  // :369-373
                const float w = ((rx * w0 + ry * w1) + rz * w2) + bb;                            // fc_1 with bias (:426-430)
                const float fk = (real && cok) ? f[(size_t)id * C + c] : 0.f;
                if (!BWD) acc += w * fk;                                                         // :457-464
                else {
                    if (real && cok && gf) unsafeAtomicAdd(gf + (size_t)id * C + c, g * w);
                    const float gw = g * fk;
                    a0 += gw * rx; a1 += gw * ry; a2 += gw * rz; ab += gw;
                }
            }
            if (!BWD && cok) out[(size_t)p * C + c] = acc / nn;
        }
        if (BWD && cok) {
            if (gfcw) { unsafeAtomicAdd(gfcw + c, a0); unsafeAtomicAdd(gfcw + C + c, a1); unsafeAtomicAdd(gfcw + 2 * C + c, a2); }
            if (gfcb) unsafeAtomicAdd(gfcb + c, ab);
        }
    }
}

// Forward, C % 4 == 0: one lane = 4 consecutive channels of one point (16-byte row segments), a point's L = C/4 lanes sit next to each other so a
// neighbour's feature row is read as one contiguous burst, U neighbours are in flight per lane, and a workgroup takes tpb = 256 / L points per trip.
// Trips are dealt to the XCDs in contiguous eighths of the processing sequence (`order`, or the row order: the pyramid's points leave the grid
// subsampling sorted by voxel key), so a support row is pulled into ONE XCD's L2 instead of all eight (round 2: plain grid-stride over n * C/4).
// The wave-per-point kernel above wastes the lanes past C (C = 72 runs a second pass with 8 of 64 lanes) and has one row in flight.
template <int U>
__global__ __launch_bounds__(256) void adaptive_weight_fwd_v4(unsigned n, int n0, int K, int C4, int c4_0, int L, const float* __restrict__ q, const float* __restrict__ s,
                                                              const int* __restrict__ idx, const float4* __restrict__ f, float inv_radius,
                                                              const float4* __restrict__ fcw, const float4* __restrict__ fcb,
                                                              const int* __restrict__ padding_num, int reduction_mean, const int* __restrict__ order,
                                                              float4* __restrict__ out)
{
    const int pad = reduction_mean ? *padding_num : 0;
    const int tpb = 256 / L;
    const int ts = threadIdx.x / L, cl = threadIdx.x - ts * L;
    if (ts >= tpb) return;
    const int cq = c4_0 + cl;
    const float4 w0 = fcw[cq], w1 = fcw[C4 + cq], w2 = fcw[2 * C4 + cq], bb = fcb[cq];
    const unsigned ntrips = (n + tpb - 1) / tpb;
    const unsigned vend = 8 * cbl_xcd_per(ntrips);
    for (unsigned v = blockIdx.x; v < vend; v += gridDim.x) {
        const unsigned tr = cbl_xcd_slot(v, ntrips) * tpb + ts;
        if (tr >= n) continue;
        const int p = order ? order[tr] : (int)tr;
        const float qx = q[3 * (size_t)p], qy = q[3 * (size_t)p + 1], qz = q[3 * (size_t)p + 2];
        const int* __restrict__ row = idx + (size_t)p * K;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int cnt = 0;
        // the ids of the NEXT batch are requested before this batch's rows: id -> row is a chain of two memory round trips per batch, and with
        // the ids one batch ahead only the first batch pays both
        int idn[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const int v_ = row[min(u, K - 1)]; idn[u] = (u < K) ? v_ : n0; }      // clamped, unconditional loads
        for (int k0 = 0; k0 < K; k0 += U) {
            int id[U]; float rx[U], ry[U], rz[U]; float4 fk[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                id[u] = idn[u];
                cnt += (k0 + u < K && id[u] < pad) ? 1 : 0;
            }
#pragma unroll
            for (int u = 0; u < U; u++) { const int v_ = row[min(k0 + U + u, K - 1)]; idn[u] = (k0 + U + u < K) ? v_ : n0; }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const bool real = id[u] >= 0 && id[u] < n0;
                const int ic = real ? id[u] : 0;
                fk[u] = f[(size_t)ic * C4 + cq];
                rx[u] = s[3 * (size_t)ic]; ry[u] = s[3 * (size_t)ic + 1]; rz[u] = s[3 * (size_t)ic + 2];
                if (!real) { fk[u] = make_float4(0.f, 0.f, 0.f, 0.f); rx[u] = ry[u] = rz[u] = 0.f; }       // shadow row / point (:360-370)
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (k0 + u < K) {
                    const float x = (rx[u] - qx) * inv_radius, y = (ry[u] - qy) * inv_radius, z = (rz[u] - qz) * inv_radius;    // :369-373 (one rounded reciprocal instead of three divisions per pair: within 1 ulp of them)
                    acc.x += (((x * w0.x + y * w1.x) + z * w2.x) + bb.x) * fk[u].x;                      // fc_1 with bias (:426-430), :457-464
                    acc.y += (((x * w0.y + y * w1.y) + z * w2.y) + bb.y) * fk[u].y;
                    acc.z += (((x * w0.z + y * w1.z) + z * w2.z) + bb.z) * fk[u].z;
                    acc.w += (((x * w0.w + y * w1.w) + z * w2.w) + bb.w) * fk[u].w;
                }
            }
        }
        const float nn = reduction_mean ? (float)cnt + 1e-5f : 1.f;
        out[(size_t)p * C4 + cq] = make_float4(acc.x / nn, acc.y / nn, acc.z / nn, acc.w / nn);
    }
}

// v5: the v4 kernel measured VALU-issue bound, not memory bound (wave-instructions x 4 clk / SIMD = ~95 of its 121 us at N = 200 000, C = 72; more rows
// in flight made it slower): ~40 vector instructions per (neighbour, four channels), two thirds of them overhead every lane of a point repeats (ids,
// addresses, the offset vector, shadow selects).  Here a lane owns CPL float4 columns (8 or 12 channels), so that overhead is paid once per 8 / 12
// channels; the fully connected layer moves out of the loop — out[p,c] = w0[c] S0 + w1[c] S1 + w2[c] S2 + b[c] S3 with
// S_a[p,c] = sum_k r_a(p,k) f[nbr_k, c], r = (offset / radius, 1): three multiply-adds and one add per (neighbour, channel), written as fmaf (the
// translation unit is compiled with -ffp-contract=off) — and a shadow neighbour is a zero multiplier instead of selects on the row.
template <int CPL, int U>
__global__ __launch_bounds__(256) void adaptive_weight_fwd_v5(unsigned n, int n0, int K, int C4, int c4_0, int L, const float* __restrict__ q, const float* __restrict__ s,
                                                              const int* __restrict__ idx, const float4* __restrict__ f, float inv_radius,
                                                              const float4* __restrict__ fcw, const float4* __restrict__ fcb,
                                                              const int* __restrict__ padding_num, int reduction_mean, const int* __restrict__ order,
                                                              float4* __restrict__ out)
{
    const int pad = reduction_mean ? *padding_num : 0;
    const int tpb = 256 / L;
    const int ts = threadIdx.x / L, cl = threadIdx.x - ts * L;
    if (ts >= tpb) return;
    const int cq = c4_0 + cl * CPL;                                    // this lane's first float4 column
    const unsigned ntrips = (n + tpb - 1) / tpb;
    const unsigned vend = 8 * cbl_xcd_per(ntrips);
    for (unsigned v = blockIdx.x; v < vend; v += gridDim.x) {
        const unsigned tr = cbl_xcd_slot(v, ntrips) * tpb + ts;
        if (tr >= n) continue;
        const int p = order ? order[tr] : (int)tr;
        const float qx = q[3 * (size_t)p], qy = q[3 * (size_t)p + 1], qz = q[3 * (size_t)p + 2];
        const int* __restrict__ row = idx + (size_t)p * K;
        float4 S0[CPL], S1[CPL], S2[CPL], S3[CPL];
#pragma unroll
        for (int j = 0; j < CPL; j++) S0[j] = S1[j] = S2[j] = S3[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        int cnt = 0;
        int idn[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const int v_ = row[min(u, K - 1)]; idn[u] = (u < K) ? v_ : n0; }      // ids one batch ahead (see v4)
        for (int k0 = 0; k0 < K; k0 += U) {
            int id[U]; float rx[U], ry[U], rz[U]; float4 fk[U][CPL];
#pragma unroll
            for (int u = 0; u < U; u++) {
                id[u] = idn[u];
                cnt += (k0 + u < K && id[u] < pad) ? 1 : 0;
            }
#pragma unroll
            for (int u = 0; u < U; u++) { const int v_ = row[min(k0 + U + u, K - 1)]; idn[u] = (k0 + U + u < K) ? v_ : n0; }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const bool real = id[u] >= 0 && id[u] < n0;
                const int ic = real ? id[u] : 0;
                const float4* __restrict__ fr = f + (size_t)ic * C4 + cq;
#pragma unroll
                for (int j = 0; j < CPL; j++) fk[u][j] = fr[j];
                rx[u] = s[3 * (size_t)ic]; ry[u] = s[3 * (size_t)ic + 1]; rz[u] = s[3 * (size_t)ic + 2];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const bool real = id[u] >= 0 && id[u] < n0;          // shadow (and past-the-row) neighbours: zero feature row = zero multipliers (:360-370)
                const float one = real ? 1.f : 0.f;
                const float x = real ? (rx[u] - qx) * inv_radius : 0.f, y = real ? (ry[u] - qy) * inv_radius : 0.f, z = real ? (rz[u] - qz) * inv_radius : 0.f;
#pragma unroll
                for (int j = 0; j < CPL; j++) {
                    const float4 fv = fk[u][j];
                    S0[j] = make_float4(fmaf(x, fv.x, S0[j].x), fmaf(x, fv.y, S0[j].y), fmaf(x, fv.z, S0[j].z), fmaf(x, fv.w, S0[j].w));
                    S1[j] = make_float4(fmaf(y, fv.x, S1[j].x), fmaf(y, fv.y, S1[j].y), fmaf(y, fv.z, S1[j].z), fmaf(y, fv.w, S1[j].w));
                    S2[j] = make_float4(fmaf(z, fv.x, S2[j].x), fmaf(z, fv.y, S2[j].y), fmaf(z, fv.z, S2[j].z), fmaf(z, fv.w, S2[j].w));
                    S3[j] = make_float4(fmaf(one, fv.x, S3[j].x), fmaf(one, fv.y, S3[j].y), fmaf(one, fv.z, S3[j].z), fmaf(one, fv.w, S3[j].w));
                }
            }
        }
        const float inv_nn = reduction_mean ? 1.0f / ((float)cnt + 1e-5f) : 1.f;
#pragma unroll
        for (int j = 0; j < CPL; j++) {
            const float4 w0 = fcw[cq + j], w1 = fcw[C4 + cq + j], w2 = fcw[2 * C4 + cq + j], bb = fcb[cq + j];
            float4 o;
            o.x = fmaf(bb.x, S3[j].x, fmaf(w2.x, S2[j].x, fmaf(w1.x, S1[j].x, w0.x * S0[j].x))) * inv_nn;
            o.y = fmaf(bb.y, S3[j].y, fmaf(w2.y, S2[j].y, fmaf(w1.y, S1[j].y, w0.y * S0[j].y))) * inv_nn;
            o.z = fmaf(bb.z, S3[j].z, fmaf(w2.z, S2[j].z, fmaf(w1.z, S1[j].z, w0.z * S0[j].z))) * inv_nn;
            o.w = fmaf(bb.w, S3[j].w, fmaf(w2.w, S2[j].w, fmaf(w1.w, S1[j].w, w0.w * S0[j].w))) * inv_nn;
            out[(size_t)p * C4 + cq + j] = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------- AdaptiveWeight backward as a gather
// Both gradients go through ONE per-target quantity (the forward transposed, as in kpconv_backward.hip):
//     S_j[a, c] = sum over the pairs p = (i, k) with nbr(i, k) = j of  r_a(i, j) * go[i, c] / nn[i],   r = ((s_j - q_i) / radius, 1)   (a = x, y, z, 1)
//     d out / d f   :  grad_f[j, c]  = sum_a fcw'[a, c] * S_j[a, c]          (fcw' = the three rows of fc_weight and fc_bias)
//     d out / d fcw':  grad_w[a, c]  = sum_j f[j, c] * S_j[a, c]
// One lane owns four consecutive channels of one target row; the lanes of a target sit next to each other (its gradient rows are read as
// contiguous bursts), a workgroup walks `tpb` targets at a time over the transposed table in its processing order.  No atomics, written not
// accumulated, deterministic: the parameter gradients are summed per lane, then over the lanes of a workgroup that hold the same channels
// (LDS), then over the workgroups (partial rows + aw_param_reduce_kernel).  Round 2 scattered d out / d f with one float atomic per (pair,
// channel): 0.9 - 2.6 ms per layer of the ConvNet at N = 200 000 (profiles/r03_bench_convnet_baseline.json).
__global__ __launch_bounds__(256) void aw_inv_count_kernel(int n, int K, const int* __restrict__ idx, const int* __restrict__ padding_num,
                                                           int reduction_mean, float* __restrict__ inv_nn)
{
    const int pad = reduction_mean ? *padding_num : 0;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) {
        int cnt = 0;
        if (reduction_mean)
            for (int k = 0; k < K; k++) cnt += idx[(size_t)p * K + k] < pad ? 1 : 0;
        inv_nn[p] = reduction_mean ? 1.0f / ((float)cnt + 1e-5f) : 1.0f;       // :466-470
    }
}

// L = lanes per target in this launch (a chunk of at most 256 float4 columns starting at column c4_0), tpb = 256 / L targets per trip.
// partial: (gridDim.x, 4, C) per-workgroup sums of (grad_fcw rows 0..2, grad_fcb).
// A lane owns CPL float4 columns of one target (8 or 12 channels: the pair id, its source row address, the offset vector and the 1 / nn factor are paid
// once per 8 / 12 channels — the kernel is VALU-issue bound like the forward), multiply-adds written as fmaf (-ffp-contract=off).
template <bool GF, bool GP, int UB, int CPL>
__global__ __launch_bounds__(256) void aw_bwd_csr_kernel(unsigned n0, int C4, int c4_0, int L, CblFastDiv dvK, const float* __restrict__ q, const float* __restrict__ s,
                                                        const float4* __restrict__ f, float inv_radius, const float4* __restrict__ fcw, const float4* __restrict__ fcb,
                                                        const float* __restrict__ inv_nn, const float4* __restrict__ go,
                                                        const int* __restrict__ order, const int* __restrict__ inv_start, const int* __restrict__ inv_src,
                                                        float4* __restrict__ gf, float* __restrict__ partial)
{
    __shared__ float4 red[4][256];
    const int tpb = 256 / L;
    const int ts = threadIdx.x / L, cl = threadIdx.x - ts * L;       // target slot of the trip, lane within the target
    const bool on = ts < tpb;
    const int cq = c4_0 + cl * CPL;                                  // this lane's first float4 column
    float4 A[4][CPL];                                                // parameter gradients of this lane's channels
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int c = 0; c < CPL; c++) A[a][c] = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned ntrips = (n0 + tpb - 1) / tpb;
    const unsigned vend = 8 * cbl_xcd_per(ntrips);
    for (unsigned v = blockIdx.x; v < vend; v += gridDim.x) {
        const unsigned tr = cbl_xcd_slot(v, ntrips) * tpb + ts;
        if (!on || tr >= n0) continue;
        const int j = order ? order[tr] : (int)tr;
        const int e0 = inv_start[tr], e1 = inv_start[tr + 1];
        const float sx = s[3 * (size_t)j], sy = s[3 * (size_t)j + 1], sz = s[3 * (size_t)j + 2];
        float4 S0[CPL], S1[CPL], S2[CPL], S3[CPL];
#pragma unroll
        for (int c = 0; c < CPL; c++) S0[c] = S1[c] = S2[c] = S3[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e1 > e0) {
            // UB pairs in flight per lane; the pair ids of the NEXT batch are requested before this batch's rows (pair -> source row is a chain
            // of two round trips, the ids one batch ahead leave one)
            int pn[UB];
#pragma unroll
            for (int u = 0; u < UB; u++) pn[u] = inv_src[min(e0 + u, e1 - 1)];
            for (int e = e0; e < e1; e += UB) {
                int pi[UB]; float4 g[UB][CPL]; float rx[UB], ry[UB], rz[UB], sc[UB];
#pragma unroll
                for (int u = 0; u < UB; u++) pi[u] = (int)cbl_fastdiv((unsigned)pn[u], dvK);
#pragma unroll
                for (int u = 0; u < UB; u++) pn[u] = inv_src[min(e + UB + u, e1 - 1)];
#pragma unroll
                for (int u = 0; u < UB; u++) {
                    const float4* __restrict__ gr = go + (size_t)pi[u] * C4 + cq;
#pragma unroll
                    for (int c = 0; c < CPL; c++) g[u][c] = gr[c];
                    rx[u] = q[3 * (size_t)pi[u]]; ry[u] = q[3 * (size_t)pi[u] + 1]; rz[u] = q[3 * (size_t)pi[u] + 2];
                    sc[u] = inv_nn[pi[u]];
                }
#pragma unroll
                for (int u = 0; u < UB; u++) {
                    const bool in = e + u < e1;                       // (uniform over the lanes of a target); past the list: zero multipliers
                    const float w = in ? sc[u] : 0.f;
                    const float x = (sx - rx[u]) * inv_radius * w, y = (sy - ry[u]) * inv_radius * w, z = (sz - rz[u]) * inv_radius * w;
#pragma unroll
                    for (int c = 0; c < CPL; c++) {
                        const float4 gv = g[u][c];
                        S0[c] = make_float4(fmaf(x, gv.x, S0[c].x), fmaf(x, gv.y, S0[c].y), fmaf(x, gv.z, S0[c].z), fmaf(x, gv.w, S0[c].w));
                        S1[c] = make_float4(fmaf(y, gv.x, S1[c].x), fmaf(y, gv.y, S1[c].y), fmaf(y, gv.z, S1[c].z), fmaf(y, gv.w, S1[c].w));
                        S2[c] = make_float4(fmaf(z, gv.x, S2[c].x), fmaf(z, gv.y, S2[c].y), fmaf(z, gv.z, S2[c].z), fmaf(z, gv.w, S2[c].w));
                        S3[c] = make_float4(fmaf(w, gv.x, S3[c].x), fmaf(w, gv.y, S3[c].y), fmaf(w, gv.z, S3[c].z), fmaf(w, gv.w, S3[c].w));
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            if (GF) {
                const float4 w0 = fcw[cq + c], w1 = fcw[C4 + cq + c], w2 = fcw[2 * C4 + cq + c], bb = fcb[cq + c];
                gf[(size_t)j * C4 + cq + c] = make_float4(fmaf(bb.x, S3[c].x, fmaf(w2.x, S2[c].x, fmaf(w1.x, S1[c].x, w0.x * S0[c].x))),
                                                          fmaf(bb.y, S3[c].y, fmaf(w2.y, S2[c].y, fmaf(w1.y, S1[c].y, w0.y * S0[c].y))),
                                                          fmaf(bb.z, S3[c].z, fmaf(w2.z, S2[c].z, fmaf(w1.z, S1[c].z, w0.z * S0[c].z))),
                                                          fmaf(bb.w, S3[c].w, fmaf(w2.w, S2[c].w, fmaf(w1.w, S1[c].w, w0.w * S0[c].w))));
            }
            if (GP) {
                const float4 fj = f[(size_t)j * C4 + cq + c];
                A[0][c] = make_float4(fmaf(fj.x, S0[c].x, A[0][c].x), fmaf(fj.y, S0[c].y, A[0][c].y), fmaf(fj.z, S0[c].z, A[0][c].z), fmaf(fj.w, S0[c].w, A[0][c].w));
                A[1][c] = make_float4(fmaf(fj.x, S1[c].x, A[1][c].x), fmaf(fj.y, S1[c].y, A[1][c].y), fmaf(fj.z, S1[c].z, A[1][c].z), fmaf(fj.w, S1[c].w, A[1][c].w));
                A[2][c] = make_float4(fmaf(fj.x, S2[c].x, A[2][c].x), fmaf(fj.y, S2[c].y, A[2][c].y), fmaf(fj.z, S2[c].z, A[2][c].z), fmaf(fj.w, S2[c].w, A[2][c].w));
                A[3][c] = make_float4(fmaf(fj.x, S3[c].x, A[3][c].x), fmaf(fj.y, S3[c].y, A[3][c].y), fmaf(fj.z, S3[c].z, A[3][c].z), fmaf(fj.w, S3[c].w, A[3][c].w));
            }
        }
    }
    if (GP) {
        // the tpb lanes that hold the same channels, in slot order (deterministic), then one partial row block per workgroup; one column set at a time
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            __syncthreads();
            red[0][threadIdx.x] = A[0][c]; red[1][threadIdx.x] = A[1][c]; red[2][threadIdx.x] = A[2][c]; red[3][threadIdx.x] = A[3][c];
            __syncthreads();
            if (threadIdx.x < L) {
                for (int a = 0; a < 4; a++) {
                    float4 sum = red[a][threadIdx.x];
                    for (int t = 1; t < tpb; t++) { const float4 o = red[a][t * L + threadIdx.x]; sum.x += o.x; sum.y += o.y; sum.z += o.z; sum.w += o.w; }
                    reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * 4 + a) * (size_t)(4 * C4))[c4_0 + threadIdx.x * CPL + c] = sum;
                }
            }
        }
    }
}

// grad_fcw / grad_fcb [e] = sum over the workgroups' partial rows, in a fixed order: 64 threads per element, each a 64th of the workgroups (16 threads
// per element in a 256-thread workgroup left 4 C / 16 = 18 workgroups at C = 72 walking 128 rows each: 40 us)
constexpr int AWR_GROUPS = 64;
__global__ __launch_bounds__(16 * AWR_GROUPS) void aw_param_reduce_kernel(int nblk, int C, const float* __restrict__ partial, float* __restrict__ gfcw,
                                                                          float* __restrict__ gfcb)
{
    __shared__ float part[AWR_GROUPS][16];
    const int total = 4 * C;
    const int el = threadIdx.x & 15, pt = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    const int per = (nblk + AWR_GROUPS - 1) / AWR_GROUPS, b0 = pt * per, b1 = min(nblk, b0 + per);
    float acc = 0.f;
    if (e < total) {
        int b = b0;
        for (; b + 4 <= b1; b += 4) {                                 // four rows in flight
            const float v0 = partial[(size_t)b * total + e], v1 = partial[(size_t)(b + 1) * total + e], v2 = partial[(size_t)(b + 2) * total + e],
                        v3 = partial[(size_t)(b + 3) * total + e];
            acc += v0; acc += v1; acc += v2; acc += v3;
        }
        for (; b < b1; b++) acc += partial[(size_t)b * total + e];
    }
    part[pt][el] = acc;
    __syncthreads();
    if (pt == 0 && e < total) {
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < AWR_GROUPS; k++) sum += part[k][el];
        if (e < 3 * C) { if (gfcw) gfcw[e] = sum; } else if (gfcb) gfcb[e - 3 * C] = sum;
    }
}

// ---------------------------------------------------------------------------------------------- index pooling
__global__ __launch_bounds__(256) void column_min_kernel(int n, int d, const float* __restrict__ x, unsigned* __restrict__ keymin)
{
    // grid.y strides rows, threads cover columns; order-preserving integer keys so that atomicMin works on floats
    for (int c = blockIdx.x * 256 + threadIdx.x; c < d; c += gridDim.x * 256) {
        float m = INFINITY;
        for (int r = blockIdx.y; r < n; r += gridDim.y) m = fminf(m, x[(size_t)r * d + c]);
        const unsigned u = __float_as_uint(m);
        atomicMin(keymin + c, (u & 0x80000000u) ? ~u : (u | 0x80000000u));
    }
}
__global__ __launch_bounds__(256) void ind_max_pool_kernel(int n1, int n2, int k, int d, const float* __restrict__ x, const int* __restrict__ inds,
                                                           const unsigned* __restrict__ keymin, float* __restrict__ out)
{
    const long long total = (long long)n2 * d;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / d; const int c = (int)(e - r * d);
        const unsigned key = keymin[c];
        const float shadow = __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
        float m = -INFINITY;
        for (int j = 0; j < k; j++) {
            const int id = inds[r * k + j];
            m = fmaxf(m, (id >= 0 && id < n1) ? x[(size_t)id * d + c] : shadow);
        }
        out[e] = m;
    }
}
// column minima for d % 4 == 0 with every lane busy (the kernel above leaves 184 of 256 lanes idle at d = 72): lane = 4 columns, 256 / L rows per step,
// running minima in registers, the row slots of a workgroup folded in LDS, four atomicMin per lane of slot 0
__global__ __launch_bounds__(256) void column_min_v4_kernel(unsigned n, int D4, int c4_0, int L, const float4* __restrict__ x, unsigned* __restrict__ keymin)
{
    __shared__ float4 red[256];
    const int tpb = 256 / L;
    const int ts = threadIdx.x / L, cl = threadIdx.x - ts * L;
    const bool on = ts < tpb;
    const int cq = c4_0 + cl;
    float4 m = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
    if (on)
        for (unsigned r = blockIdx.x * tpb + ts; r < n; r += gridDim.x * tpb) {
            const float4 v = x[(size_t)r * D4 + cq];
            m.x = fminf(m.x, v.x); m.y = fminf(m.y, v.y); m.z = fminf(m.z, v.z); m.w = fminf(m.w, v.w);
        }
    red[threadIdx.x] = m;
    __syncthreads();
    if (threadIdx.x < L) {
        for (int t = 1; t < tpb; t++) { const float4 o = red[t * L + threadIdx.x]; m.x = fminf(m.x, o.x); m.y = fminf(m.y, o.y); m.z = fminf(m.z, o.z); m.w = fminf(m.w, o.w); }
        const float mv[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
        for (int j = 0; j < 4; j++) { const unsigned u = __float_as_uint(mv[j]); atomicMin(keymin + 4 * cq + j, (u & 0x80000000u) ? ~u : (u | 0x80000000u)); }
    }
}

// d % 4 == 0: lane = 4 channels of one pooled row, L = d/4 lanes per row (a chunk of at most 256 columns), 256 / L rows per trip dealt to the XCDs in contiguous
// eighths, U rows in flight with their ids one batch ahead (the kernel above: one channel per lane, a 64-bit division per element, one row in flight)
template <int U>
__global__ __launch_bounds__(256) void ind_max_pool_v4_kernel(unsigned n2, int n1, int k, int D4, int c4_0, int L, const float4* __restrict__ x,
                                                              const int* __restrict__ inds, const unsigned* __restrict__ keymin, float4* __restrict__ out)
{
    const int tpb = 256 / L;
    const int ts = threadIdx.x / L, cl = threadIdx.x - ts * L;
    if (ts >= tpb) return;
    const int cq = c4_0 + cl;
    float sh[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { const unsigned key = keymin[4 * cq + j]; sh[j] = __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key); }
    const unsigned ntrips = (n2 + tpb - 1) / tpb;
    const unsigned vend = 8 * cbl_xcd_per(ntrips);
    for (unsigned v = blockIdx.x; v < vend; v += gridDim.x) {
        const unsigned r = cbl_xcd_slot(v, ntrips) * tpb + ts;
        if (r >= n2) continue;
        const int* __restrict__ row = inds + (size_t)r * k;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        int idn[U];
#pragma unroll
        for (int u = 0; u < U; u++) idn[u] = row[min(u, k - 1)];
        for (int k0 = 0; k0 < k; k0 += U) {
            int id[U]; float4 xr[U];
#pragma unroll
            for (int u = 0; u < U; u++) id[u] = idn[u];
#pragma unroll
            for (int u = 0; u < U; u++) idn[u] = row[min(k0 + U + u, k - 1)];        // past the row: the last entry again (max is idempotent)
#pragma unroll
            for (int u = 0; u < U; u++) { const bool real = id[u] >= 0 && id[u] < n1; xr[u] = x[(size_t)(real ? id[u] : 0) * D4 + cq]; }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const bool real = id[u] >= 0 && id[u] < n1;
                m.x = fmaxf(m.x, real ? xr[u].x : sh[0]); m.y = fmaxf(m.y, real ? xr[u].y : sh[1]);
                m.z = fmaxf(m.z, real ? xr[u].z : sh[2]); m.w = fmaxf(m.w, real ? xr[u].w : sh[3]);
            }
        }
        out[(size_t)r * D4 + cq] = m;
    }
}
__global__ __launch_bounds__(256) void ind_closest_pool_kernel(int n1, int n2, int k, int d, const float* __restrict__ x, const int* __restrict__ inds,
                                                               float* __restrict__ out)
{
    const long long total = (long long)n2 * d;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / d; const int c = (int)(e - r * d);
        const int id = inds[r * k];
        out[e] = (id >= 0 && id < n1) ? x[(size_t)id * d + c] : 0.f;
    }
}
__global__ void fill_u32_kernel(int n, unsigned v, unsigned* __restrict__ p) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }

// workgroups (256 lanes) of `kernel` that are resident on the device at once: persistent kernels launch no more than that — a fixed
// 2048-workgroup grid leaves a second, partly empty round wherever the kernel's registers allow fewer than 8 waves per SIMD
template <class F>
inline unsigned resident_workgroups(F kernel, int& cache)
{
    if (!cache) {
        int per_cu = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kernel), 256, 0) != hipSuccess || per_cu <= 0) per_cu = 4;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cache = per_cu * cus;
    }
    return (unsigned)cache;
}

inline unsigned persistent_grid(int n) { const long long waves = n; long long blocks = (waves + 3) / 4; if (blocks > 256 * 8) blocks = 256 * 8; if (blocks < 1) blocks = 1; return (unsigned)blocks; }

}  // namespace

static int kpconv_forward_impl(int n, int n0, int K, int C, int KP, const float* query_points, const float* support_points, const int* neighbors_indices,
                               const float* features, const float* kernel_points, const float* kernel_weights, float extent, int influence, int closest,
                               const int* order, float* out, void* stream)
{
    if (n < 0 || n0 < 0 || K <= 0 || C <= 0 || KP <= 0 || KP > 16 || !(extent > 0.f) || influence < 0 || influence > 1) return CBL_ERR_BAD_ARG;
    if (n == 0) return CBL_OK;
    if (!query_points || !support_points || !neighbors_indices || !features || !kernel_points || !kernel_weights || !out) return CBL_ERR_BAD_ARG;
    hipStream_t st = cbl_stream(stream);
    // persistent waves: exactly as many workgroups as are resident at once (the kernel's registers allow 5 waves per SIMD, not 8: with the
    // fixed 2048-workgroup grid 768 of them ran as a second, mostly idle round), each walking its share of the points
    static int resident[2] = {0, 0};
    const bool vec = C % 4 == 0 && cbl_host_aligned16(features) && cbl_host_aligned16(out);
    static int resident64[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool c64 = vec && C <= 64 && cbl_host_aligned16(kernel_weights) && (unsigned long long)n0 * C * 4ull < (1ull << 32);
#define CBL_KPF(CL, LI, ON) do { const dim3 grid(cbl_round_up8(min(persistent_grid(n), resident_workgroups(&kpconv_fwd_c64_kernel<CL, LI, ON>, resident64[4 * CL + 2 * LI + ON])))); \
        hipLaunchKernelGGL((kpconv_fwd_c64_kernel<CL, LI, ON>), grid, dim3(256), 0, st, n, n0, K, C, KP, query_points, support_points, neighbors_indices, features, \
                           kernel_points, kernel_weights, extent, order, out); return cbl_status(); } while (0)
#define CBL_KPF2(CL, LI) do { if (K <= 16) CBL_KPF(CL, LI, true); else CBL_KPF(CL, LI, false); } while (0)
    if (c64) {
        if (closest) { if (influence) CBL_KPF2(true, true); else CBL_KPF2(true, false); }
        else         { if (influence) CBL_KPF2(false, true); else CBL_KPF2(false, false); }
    }
#undef CBL_KPF2
#undef CBL_KPF
    const unsigned res = vec ? resident_workgroups(&kpconv_fwd_kernel<true>, resident[1]) : resident_workgroups(&kpconv_fwd_kernel<false>, resident[0]);
    const dim3 grid(cbl_round_up8(min(persistent_grid(n), res))), block(256);
    if (vec)
        hipLaunchKernelGGL(kpconv_fwd_kernel<true>, grid, block, 0, st, n, n0, K, C, KP, query_points, support_points, neighbors_indices, features, kernel_points, kernel_weights, extent, influence, closest, order, out);
    else
        hipLaunchKernelGGL(kpconv_fwd_kernel<false>, grid, block, 0, st, n, n0, K, C, KP, query_points, support_points, neighbors_indices, features, kernel_points, kernel_weights, extent, influence, closest, order, out);
    return cbl_status();
}

CBL_EXPORT int cbl_kpconv_forward(int n, int n0, int K, int C, int KP, const float* query_points, const float* support_points, const int* neighbors_indices,
                                  const float* features, const float* kernel_points, const float* kernel_weights, float extent, int influence, int closest,
                                  float* out, void* stream)
{
    return kpconv_forward_impl(n, n0, K, C, KP, query_points, support_points, neighbors_indices, features, kernel_points, kernel_weights, extent, influence, closest,
                               nullptr, out, stream);
}

CBL_EXPORT int cbl_kpconv_forward_ordered(int n, int n0, int K, int C, int KP, const float* query_points, const float* support_points, const int* neighbors_indices,
                                          const float* features, const float* kernel_points, const float* kernel_weights, float extent, int influence, int closest,
                                          const int* order, float* out, void* stream)
{
    return kpconv_forward_impl(n, n0, K, C, KP, query_points, support_points, neighbors_indices, features, kernel_points, kernel_weights, extent, influence, closest,
                               order, out, stream);
}

CBL_EXPORT int cbl_kpconv_backward(int n, int n0, int K, int C, int KP, const float* query_points, const float* support_points, const int* neighbors_indices,
                                   const float* features, const float* kernel_points, const float* kernel_weights, float extent, int influence, int closest,
                                   const float* grad_out, float* grad_features, float* grad_kernel_weights, void* stream)
{
    if (n < 0 || n0 < 0 || K <= 0 || K > KPB_MAXK || C <= 0 || KP <= 0 || KP > KPB_MAXKP || !(extent > 0.f) || influence < 0 || influence > 1) return CBL_ERR_BAD_ARG;
    if (n == 0) return CBL_OK;
    if (!query_points || !support_points || !neighbors_indices || !features || !kernel_points || !kernel_weights || !grad_out) return CBL_ERR_BAD_ARG;
    static int res_bwd = 0;
    hipLaunchKernelGGL(kpconv_bwd_kernel, dim3(min(persistent_grid(n), resident_workgroups(&kpconv_bwd_kernel, res_bwd))), dim3(256), 0, cbl_stream(stream), n, n0, K, C, KP, query_points, support_points, neighbors_indices,
                       features, kernel_points, kernel_weights, extent, influence, closest, grad_out, grad_features, grad_kernel_weights);
    return cbl_status();
}

CBL_EXPORT int cbl_index_max(long long total, const int* idx, int* out_max, void* stream)
{
    if (total <= 0 || !idx || !out_max) return CBL_ERR_BAD_ARG;
    hipStream_t st = cbl_stream(stream);
    hipLaunchKernelGGL(fill_u32_kernel, dim3(1), dim3(64), 0, st, 1, 0x80000000u, reinterpret_cast<unsigned*>(out_max));   // INT_MIN
    const int vec = (reinterpret_cast<uintptr_t>(idx) & 15) == 0;
    hipLaunchKernelGGL(index_max_kernel, dim3(cbl_grid_for((total + 3) / 4, 256, 512)), dim3(256), 0, st, total, vec, idx, out_max);
    return cbl_status();
}


// float4 columns per lane (measured at N = 200 000, C = 72 .. 1152, profiles/r03_aw_lane_width_sweep.md): the forward is quickest with 3 where C / 4
// divides by 3 (the ConvNet's 72 * 2^l), the backward with 2; two neighbours / pairs in flight per lane in both (4 and 8 cost occupancy: slower)
static int aw_cpl_for(int c4) { return c4 % 3 == 0 ? 3 : (c4 % 2 == 0 ? 2 : 1); }
static int aw_cpl_bwd(int c4) { return c4 % 2 == 0 ? 2 : (c4 % 3 == 0 ? 3 : 1); }

static int adaptive_weight_forward_impl(int n, int n0, int K, int C, const float* query_points, const float* support_points, const int* neighbors_indices,
                                        const float* features, float radius, const float* fc_weight, const float* fc_bias, const int* padding_num,
                                        int reduction_mean, const int* order, float* out, void* stream)
{
    if (n < 0 || n0 < 0 || K <= 0 || C <= 0 || !(radius > 0.f)) return CBL_ERR_BAD_ARG;
    if (n == 0) return CBL_OK;
    if (!query_points || !support_points || !neighbors_indices || !features || !fc_weight || !fc_bias || !out || (reduction_mean && !padding_num)) return CBL_ERR_BAD_ARG;
    const bool vec = (C % 4 == 0) && ((((uintptr_t)features | (uintptr_t)fc_weight | (uintptr_t)fc_bias | (uintptr_t)out) & 15) == 0);
    const int C4v = C / 4;
    if (vec && aw_cpl_for(C4v) > 1) {
        const int cpl = aw_cpl_for(C4v);
        const int Lall = C4v / cpl, chunks = (Lall + 255) / 256, Lmax = (Lall + chunks - 1) / chunks;
        for (int l0 = 0; l0 < Lall; l0 += Lmax) {
            const int L = min(Lmax, Lall - l0);
            const unsigned g = min(cbl_round_up8(cbl_div_up(n, 256 / L)), 8192u);
#define CBL_AWF5(CPL_, U_) hipLaunchKernelGGL((adaptive_weight_fwd_v5<CPL_, U_>), dim3(g), dim3(256), 0, cbl_stream(stream), (unsigned)n, n0, K, C4v, l0 * cpl, L, \
                               query_points, support_points, neighbors_indices, reinterpret_cast<const float4*>(features), 1.0f / radius, \
                               reinterpret_cast<const float4*>(fc_weight), reinterpret_cast<const float4*>(fc_bias), padding_num, reduction_mean, order, \
                               reinterpret_cast<float4*>(out))
            if (cpl == 3) CBL_AWF5(3, 2); else CBL_AWF5(2, 2);
#undef CBL_AWF5
        }
    }
    else if (vec) {
        const int C4 = C / 4, chunks = (C4 + 255) / 256, Lmax = (C4 + chunks - 1) / chunks;
        for (int c4_0 = 0; c4_0 < C4; c4_0 += Lmax) {
            const int L = min(Lmax, C4 - c4_0);
            const unsigned g = min(cbl_round_up8(cbl_div_up(n, 256 / L)), 8192u);
#define CBL_AWF(U_) hipLaunchKernelGGL(adaptive_weight_fwd_v4<U_>, dim3(g), dim3(256), 0, cbl_stream(stream), (unsigned)n, n0, K, C4, c4_0, L, \
                               query_points, support_points, neighbors_indices, reinterpret_cast<const float4*>(features), 1.0f / radius, \
                               reinterpret_cast<const float4*>(fc_weight), reinterpret_cast<const float4*>(fc_bias), padding_num, reduction_mean, order, \
                               reinterpret_cast<float4*>(out))
            CBL_AWF(2);
#undef CBL_AWF
        }
    }
    else {
        static int res_aw = 0;
        hipLaunchKernelGGL(adaptive_weight_kernel<false>, dim3(min(persistent_grid(n), resident_workgroups(&adaptive_weight_kernel<false>, res_aw))), dim3(256), 0, cbl_stream(stream), n, n0, K, C, query_points, support_points,
                           neighbors_indices, features, radius, fc_weight, fc_bias, padding_num, reduction_mean, out, nullptr, nullptr, nullptr, nullptr);
    }
    return cbl_status();
}

CBL_EXPORT int cbl_adaptive_weight_forward(int n, int n0, int K, int C, const float* query_points, const float* support_points, const int* neighbors_indices,
                                           const float* features, float radius, const float* fc_weight, const float* fc_bias, const int* padding_num,
                                           int reduction_mean, float* out, void* stream)
{
    return adaptive_weight_forward_impl(n, n0, K, C, query_points, support_points, neighbors_indices, features, radius, fc_weight, fc_bias, padding_num,
                                        reduction_mean, nullptr, out, stream);
}

CBL_EXPORT int cbl_adaptive_weight_forward_ordered(int n, int n0, int K, int C, const float* query_points, const float* support_points, const int* neighbors_indices,
                                                   const float* features, float radius, const float* fc_weight, const float* fc_bias, const int* padding_num,
                                                   int reduction_mean, const int* order, float* out, void* stream)
{
    return adaptive_weight_forward_impl(n, n0, K, C, query_points, support_points, neighbors_indices, features, radius, fc_weight, fc_bias, padding_num,
                                        reduction_mean, order, out, stream);
}

CBL_EXPORT int cbl_adaptive_weight_backward(int n, int n0, int K, int C, const float* query_points, const float* support_points, const int* neighbors_indices,
                                            const float* features, float radius, const float* fc_weight, const float* fc_bias, const int* padding_num,
                                            int reduction_mean, const float* grad_out, float* grad_features, float* grad_fc_weight, float* grad_fc_bias, void* stream)
{
    if (n < 0 || n0 < 0 || K <= 0 || C <= 0 || !(radius > 0.f)) return CBL_ERR_BAD_ARG;
    if (n == 0) return CBL_OK;
    if (!query_points || !support_points || !neighbors_indices || !features || !fc_weight || !fc_bias || !grad_out || (reduction_mean && !padding_num)) return CBL_ERR_BAD_ARG;
    static int res_awb = 0;
    hipLaunchKernelGGL(adaptive_weight_kernel<true>, dim3(min(persistent_grid(n), resident_workgroups(&adaptive_weight_kernel<true>, res_awb))), dim3(256), 0, cbl_stream(stream), n, n0, K, C, query_points, support_points,
                       neighbors_indices, features, radius, fc_weight, fc_bias, padding_num, reduction_mean, nullptr, grad_out, grad_features, grad_fc_weight, grad_fc_bias);
    return cbl_status();
}

constexpr unsigned AW_MAX_GRID = 2048;
static unsigned aw_csr_grid(int n0, int L)
{
    const int tpb = 256 / L;
    const unsigned g = cbl_round_up8(cbl_div_up(n0, tpb));
    return g > AW_MAX_GRID ? AW_MAX_GRID : g;
}

CBL_EXPORT size_t cbl_adaptive_weight_backward_csr_workspace_bytes(int n, int n0, int C)
{
    if (n <= 0 || n0 <= 0 || C <= 0) return 0;
    return sizeof(float) * ((size_t)n + 256 + (size_t)AW_MAX_GRID * 4 * (size_t)C);           // 1 / nn per query point + per-workgroup partial rows
}

CBL_EXPORT int cbl_adaptive_weight_backward_csr(int n, int n0, int K, int C, const float* query_points, const float* support_points, const int* neighbors_indices,
                                                const float* features, float radius, const float* fc_weight, const float* fc_bias, const int* padding_num,
                                                int reduction_mean, const float* grad_out, const int* order_dst, const int* inv_start, const int* inv_src,
                                                float* grad_features, float* grad_fc_weight, float* grad_fc_bias, void* workspace, size_t workspace_bytes, void* stream)
{
    if (n < 0 || n0 < 0 || K <= 0 || C <= 0 || !(radius > 0.f)) return CBL_ERR_BAD_ARG;
    if (n == 0 || n0 == 0) return CBL_OK;
    if (!query_points || !support_points || !neighbors_indices || !features || !fc_weight || !fc_bias || !grad_out || !inv_start || !inv_src ||
        (reduction_mean && !padding_num)) return CBL_ERR_BAD_ARG;
    if (C % 4 || !cbl_host_aligned16(features) || !cbl_host_aligned16(grad_out) || !cbl_host_aligned16(fc_weight) || !cbl_host_aligned16(fc_bias) ||
        (grad_features && !cbl_host_aligned16(grad_features))) return CBL_ERR_UNSUPPORTED;
    const bool gp = grad_fc_weight || grad_fc_bias;
    if (!grad_features && !gp) return CBL_OK;
    if (!workspace || workspace_bytes < cbl_adaptive_weight_backward_csr_workspace_bytes(n, n0, C)) return CBL_ERR_WORKSPACE;
    hipStream_t st = cbl_stream(stream);
    float* inv_nn = reinterpret_cast<float*>(workspace);
    float* partial = inv_nn + (((size_t)n + 255) & ~(size_t)255);
    hipLaunchKernelGGL(aw_inv_count_kernel, dim3(cbl_grid_for(n, 256)), dim3(256), 0, st, n, K, neighbors_indices, padding_num, reduction_mean, inv_nn);
    const int C4 = C / 4;
    const CblFastDiv dv = cbl_fastdiv_make((unsigned)K);
    // lanes own cpl float4 columns; lanes in chunks of at most 256 per target; every chunk walks the table once with grid `g` (the same for all: the
    // partial rows line up)
    const int cpl = aw_cpl_bwd(C4);
    const int Lall = C4 / cpl, chunks = (Lall + 255) / 256;
    const int Lmax = (Lall + chunks - 1) / chunks;
    const unsigned g = aw_csr_grid(n0, Lmax);
    for (int l0 = 0; l0 < Lall; l0 += Lmax) {
        const int L = min(Lmax, Lall - l0);
#define CBL_AWB2(GF_, GP_, UB_, CPL_) hipLaunchKernelGGL((aw_bwd_csr_kernel<GF_, GP_, UB_, CPL_>), dim3(g), dim3(256), 0, st, (unsigned)n0, C4, l0 * cpl, L, dv, query_points, support_points, \
        reinterpret_cast<const float4*>(features), 1.0f / radius, reinterpret_cast<const float4*>(fc_weight), reinterpret_cast<const float4*>(fc_bias), inv_nn, \
        reinterpret_cast<const float4*>(grad_out), order_dst, inv_start, inv_src, reinterpret_cast<float4*>(grad_features), partial)
#define CBL_AWB(GF_, GP_) do { if (cpl == 3) CBL_AWB2(GF_, GP_, 2, 3); else if (cpl == 2) CBL_AWB2(GF_, GP_, 2, 2); else CBL_AWB2(GF_, GP_, 2, 1); } while (0)
        if (grad_features && gp) CBL_AWB(true, true); else if (grad_features) CBL_AWB(true, false); else CBL_AWB(false, true);
#undef CBL_AWB
#undef CBL_AWB2
    }
    if (gp)
        hipLaunchKernelGGL(aw_param_reduce_kernel, dim3(cbl_div_up(4 * C, 16)), dim3(16 * AWR_GROUPS), 0, st, (int)g, C, partial, grad_fc_weight, grad_fc_bias);
    return cbl_status();
}

CBL_EXPORT int cbl_ind_max_pool(int n1, int n2, int k, int d, const float* x, const int* inds, unsigned* scratch_d, float* out, void* stream)
{
    if (n1 <= 0 || n2 < 0 || k <= 0 || d <= 0) return CBL_ERR_BAD_ARG;
    if (n2 == 0) return CBL_OK;
    if (!x || !inds || !scratch_d || !out) return CBL_ERR_BAD_ARG;
    hipStream_t st = cbl_stream(stream);
    hipLaunchKernelGGL(fill_u32_kernel, dim3(cbl_div_up(d, 256)), dim3(256), 0, st, d, 0xffffffffu, scratch_d);
    const bool vec = d % 4 == 0 && ((((uintptr_t)x | (uintptr_t)out) & 15) == 0);
    if (!vec) hipLaunchKernelGGL(column_min_kernel, dim3(cbl_div_up(d, 256), (unsigned)min(n1, 512)), dim3(256), 0, st, n1, d, x, scratch_d);
    if (vec) {
        const int D4 = d / 4, chunks = (D4 + 255) / 256, Lmax = (D4 + chunks - 1) / chunks;
        for (int c4_0 = 0; c4_0 < D4 && n1 > 0; c4_0 += Lmax) {
            const int L = min(Lmax, D4 - c4_0);
            hipLaunchKernelGGL(column_min_v4_kernel, dim3(min(cbl_div_up(n1, 256 / L), 256)), dim3(256), 0, st, (unsigned)n1, D4, c4_0, L, reinterpret_cast<const float4*>(x), scratch_d);
        }
        for (int c4_0 = 0; c4_0 < D4; c4_0 += Lmax) {
            const int L = min(Lmax, D4 - c4_0);
            const unsigned g = min(cbl_round_up8(cbl_div_up(n2, 256 / L)), 8192u);
            hipLaunchKernelGGL(ind_max_pool_v4_kernel<4>, dim3(g), dim3(256), 0, st, (unsigned)n2, n1, k, D4, c4_0, L, reinterpret_cast<const float4*>(x), inds, scratch_d,
                               reinterpret_cast<float4*>(out));
        }
    }
    else hipLaunchKernelGGL(ind_max_pool_kernel, dim3(cbl_grid_for((long long)n2 * d, 256)), dim3(256), 0, st, n1, n2, k, d, x, inds, scratch_d, out);
    return cbl_status();
}

CBL_EXPORT int cbl_ind_closest_pool(int n1, int n2, int k, int d, const float* x, const int* inds, float* out, void* stream)
{
    if (n1 < 0 || n2 < 0 || k <= 0 || d <= 0) return CBL_ERR_BAD_ARG;
    if (n2 == 0) return CBL_OK;
    if (!x || !inds || !out) return CBL_ERR_BAD_ARG;
    hipLaunchKernelGGL(ind_closest_pool_kernel, dim3(cbl_grid_for((long long)n2 * d, 256)), dim3(256), 0, cbl_stream(stream), n1, n2, k, d, x, inds, out);
    return cbl_status();
}
