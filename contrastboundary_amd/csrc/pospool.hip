// a14: PosPool — parameter-free local aggregation over radius neighbourhoods.
// Replaces the TF1 op chain of PosPool  /root/reference/tensorflow/models/local_aggregation_operators.py:15-250
// (shipped config config/s3dis/pospool.yaml:20-23: position_embedding 'sin_cos', reduction 'mean'):
//     out[p, c] = reduce_k  geo[p, k, c / shared] * features[nbr(p,k), c]
// where geo is a position embedding of the neighbour's offset (support[nbr] - query[p]) / radius (:68-73) with `mid` entries,
// each shared by C / mid consecutive channels (:228-231), and the reduction is sum / mean (with the padding-count quirk
// :236-242) / max (shadow entries pushed to -65535, :243-249).  Index == n0 is the shadow neighbour: zero features, point (0,0,0).
//
// MI355X mapping.  Forward (sum / mean, C % 4 == 0): one lane = 4 consecutive channels of one point, the C/4 lanes of a point sit
// next to each other (a neighbour's feature row is one contiguous burst) and 4 neighbours are in flight per lane.  Backward and
// 'max': one wave per query point, lane = channel, neighbour ids / offsets staged in LDS once per point.  Either way the embedding
// value of (neighbour, channel) is computed in registers from per-lane constants (which monomial / direction component / sin or
// cos and its wavelength); nothing of shape (n, K, .) exists in memory (the reference materialises four such tensors).
// Bound by the gathered rows (L2 / Infinity Cache): algorithmic bytes 12n + 12n0 + 4nK + 4n0C + 4nC.
#include "cbl_common.h"

namespace {

enum { PE_ONE = 0, PE_XYZ, PE_DISTANCE, PE_EXP_D, PE_DIR_EXP_D, PE_DIR_D, PE_SIN_COS, PE_TWO_ORDER, PE_THREE_ORDER, PE_COUNT };
enum { RED_SUM = 0, RED_MEAN = 1, RED_MAX = 2 };
enum { G_ONE = 0, G_MONO, G_DIST, G_EXPD, G_DIR, G_SIN };

struct LaneGeo {            // what this lane's channel multiplies its feature with
    int kind;
    int ex, ey, ez;         // G_MONO: exponents of the normalised offset (x^ex y^ey z^ez); G_DIR / G_SIN / G_COS: ex = axis
    float dm;               // G_SIN / G_COS: wavelength divisor
    float ph;               // G_SIN: phase in turns (0.25 = the cosine: cos x = sin(x + pi/2), one code path for both halves of the embedding)
};

// `mid` of (embedding, C) as in :75-226; 0 if the reference's reshape (:229) cannot work for this C
__host__ __device__ inline int pospool_mid(int pe, int C)
{
    switch (pe) {
    case PE_ONE: case PE_DISTANCE: case PE_EXP_D: return 1;
    case PE_XYZ: return C % 3 == 0 ? 3 : 0;
    case PE_DIR_EXP_D: case PE_DIR_D: return C <= 18 ? (C % 9 == 0 ? 9 : 0) : (C % 4 == 0 ? 4 : 0);
    case PE_SIN_COS: return (C == 9 || C % 6 == 0) ? C : 0;
    case PE_TWO_ORDER: return C % 9 == 0 ? 9 : 0;
    case PE_THREE_ORDER: return C == 9 ? 9 : (C % 18 == 0 ? 18 : 0);
    }
    return 0;
}

__device__ inline LaneGeo decode_geo(int pe, int C, int c)
{
    // exponent triples of [x y z xy xz yz xx yy zz | xxx yyy zzz xxy xxz yyx yyz zzx zzy]  (:146-205)
    const unsigned char mono[18][3] = {{1,0,0},{0,1,0},{0,0,1},{1,1,0},{1,0,1},{0,1,1},{2,0,0},{0,2,0},{0,0,2},
                                       {3,0,0},{0,3,0},{0,0,3},{2,1,0},{2,0,1},{1,2,0},{0,2,1},{1,0,2},{0,1,2}};
    LaneGeo g{G_ONE, 0, 0, 0, 1.f, 0.f};
    const int mid = pospool_mid(pe, C);
    if (mid <= 0 || c >= C) return g;
    const int j = c / (C / mid);                                    // feature_map reshape [mid, shared] (:229)
    switch (pe) {
    case PE_ONE: break;
    case PE_XYZ: g.kind = G_MONO; g.ex = mono[j][0]; g.ey = mono[j][1]; g.ez = mono[j][2]; break;
    case PE_DISTANCE: g.kind = G_DIST; break;
    case PE_EXP_D: g.kind = G_EXPD; break;
    case PE_DIR_EXP_D: case PE_DIR_D: {
        // mid 9: [dir, d, dir, d, d]; mid 4: [dir, d]  (:98-117)
        const int sel = (mid == 9) ? (j < 3 ? j : j == 3 ? 3 : j < 7 ? j - 4 : 3) : j;
        if (sel < 3) { g.kind = G_DIR; g.ex = sel; } else g.kind = (pe == PE_DIR_EXP_D) ? G_EXPD : G_DIST;
        break; }
    case PE_SIN_COS:
        if (C == 9) {                                               // [sin cos](x), (y), (z), then the offset itself (:120-134)
            if (c < 6) { g.kind = G_SIN; g.ph = (c & 1) ? 0.25f : 0.f; g.ex = c >> 1; g.dm = 1.f; }
            else { g.kind = G_MONO; g.ex = mono[c - 6][0]; g.ey = mono[c - 6][1]; g.ez = mono[c - 6][2]; }
        } else {                                                    // [3, 2*feat_dim] row-major (:135-147)
            const int fd = C / 6, a = c / (2 * fd), r = c % (2 * fd), i = r < fd ? r : r - fd;
            g.kind = G_SIN; g.ph = r < fd ? 0.f : 0.25f; g.ex = a;
            g.dm = powf(1000.f, (1.0f / (float)fd) * (float)i);     // tf.pow(1.0 * wave_length, (1.0 / feat_dim) * feat_range)
        }
        break;
    case PE_TWO_ORDER: case PE_THREE_ORDER: g.kind = G_MONO; g.ex = mono[j][0]; g.ey = mono[j][1]; g.ez = mono[j][2]; break;
    }
    return g;
}

__device__ __forceinline__ float ipow(float v, int e) { return e == 0 ? 1.f : e == 1 ? v : e == 2 ? v * v : (v * v) * v; }

__device__ __forceinline__ float eval_geo(const LaneGeo& g, float rx, float ry, float rz)
{
    switch (g.kind) {
    case G_MONO: return (ipow(rx, g.ex) * ipow(ry, g.ey)) * ipow(rz, g.ez);
    case G_DIST: return sqrtf((rx * rx + ry * ry) + rz * rz);                                   // :72
    case G_EXPD: return expf(-1.0f * sqrtf((rx * rx + ry * ry) + rz * rz));                    // :90, :99
    case G_DIR: { const float d = sqrtf((rx * rx + ry * ry) + rz * rz);                         // :73
                  return (g.ex == 0 ? rx : g.ex == 1 ? ry : rz) / (d + 1e-6f); }
    // alpha = 100 (:124,:138).  v_sin_f32 on the argument in turns, reduced by v_fract: the argument reaches 100 rad, where one fp32 ulp of it is
    // already 8e-6 rad — the library sinf / cosf (a ~50-instruction path per channel and neighbour) buys nothing beyond that
    case G_SIN: return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(((100.f * (g.ex == 0 ? rx : g.ex == 1 ? ry : rz)) / g.dm) * 0.15915494309189535f + g.ph));
    }
    return 1.f;
}

constexpr int PP_KMAX = 128;      // neighbour ids / offsets of one point are staged in LDS per wave

template <bool BWD>
__global__ __launch_bounds__(256) void pospool_kernel(int n, int n0, int K, int C, const float* __restrict__ q, const float* __restrict__ s,
                                                      const int* __restrict__ idx, const float* __restrict__ f, float radius, int pe, int reduction,
                                                      const int* __restrict__ padding_num, float* __restrict__ out,
                                                      const float* __restrict__ go, float* __restrict__ gf)
{
    __shared__ int s_id[4][PP_KMAX];
    __shared__ float s_rel[4][PP_KMAX][3];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wave0 = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6), nwaves = (gridDim.x * 256) >> 6;
    const int pad = (reduction == RED_MEAN) ? *padding_num : 0;
    for (int p = wave0; p < n; p += nwaves) {
        const float qx = q[3 * p], qy = q[3 * p + 1], qz = q[3 * p + 2];
        int cnt = 0;
        for (int k = lane; k < K; k += 64) {
            const int id = idx[(size_t)p * K + k];
            const bool real = id >= 0 && id < n0;
            s_id[wv][k] = real ? id : -1;
            s_rel[wv][k][0] = ((real ? s[3 * id] : 0.f) - qx) / radius;          // shadow point = (0,0,0)  (:66-70)
            s_rel[wv][k][1] = ((real ? s[3 * id + 1] : 0.f) - qy) / radius;
            s_rel[wv][k][2] = ((real ? s[3 * id + 2] : 0.f) - qz) / radius;
            cnt += (id < pad) ? 1 : 0;
        }
        float nn = 1.f;
        if (reduction == RED_MEAN) {
            for (int sft = 32; sft >= 1; sft >>= 1) cnt += __shfl_xor(cnt, sft);
            nn = (float)cnt + 1e-5f;                                              // :238-241
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int c0 = 0; c0 < C; c0 += 64) {
            const int c = c0 + lane;
            const bool cok = c < C;
            const LaneGeo g = decode_geo(pe, C, c);
            if (reduction != RED_MAX) {
                const float gp = (BWD && cok) ? go[(size_t)p * C + c] / nn : 0.f;
                float acc = 0.f;
                for (int k = 0; k < K; k++) {
                    const int id = s_id[wv][k];                                   // wave-uniform
                    if (id < 0) continue;                                         // zero feature row: contributes nothing
                    const float ge = eval_geo(g, s_rel[wv][k][0], s_rel[wv][k][1], s_rel[wv][k][2]);
                    if (!BWD) acc += cok ? ge * f[(size_t)id * C + c] : 0.f;      // :230-235
                    else if (cok) unsafeAtomicAdd(gf + (size_t)id * C + c, gp * ge);
                }
                if (!BWD && cok) out[(size_t)p * C + c] = acc / nn;
            } else {
                // max over the K entries of geo*feature (+ -65535 on shadow entries, :243-249); gradient as tf.reduce_max's:
                // shared equally by the entries that attain the maximum
                float m = -INFINITY; int ties = 0;
                for (int k = 0; k < K; k++) {
                    const int id = s_id[wv][k];
                    const float ge = eval_geo(g, s_rel[wv][k][0], s_rel[wv][k][1], s_rel[wv][k][2]);
                    const float v = (id >= 0) ? (cok ? ge * f[(size_t)id * C + c] : 0.f) : ge * 0.f + -65535.f;
                    if (v > m) { m = v; ties = 1; } else if (v == m) ties++;
                }
                if (!BWD) { if (cok) out[(size_t)p * C + c] = m; }
                else if (cok) {
                    const float gp = go[(size_t)p * C + c] / (float)ties;
                    for (int k = 0; k < K; k++) {
                        const int id = s_id[wv][k];
                        if (id < 0) continue;
                        const float ge = eval_geo(g, s_rel[wv][k][0], s_rel[wv][k][1], s_rel[wv][k][2]);
                        if (ge * f[(size_t)id * C + c] == m) unsafeAtomicAdd(gf + (size_t)id * C + c, gp * ge);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                                          // the next point overwrites the wave's LDS rows
    }
}

// Forward for sum / mean, C % 4 == 0: lane = 4 consecutive channels of one point, L = C/4 lanes per point (a chunk of at most 256 columns), 256 / L points per
// trip, trips dealt to the XCDs in contiguous eighths — as adaptive_weight_fwd_v4.  A lane's four embedding descriptors are decoded ONCE (round 2 decoded them
// per work item of a grid-stride loop: four powf per item for 'sin_cos'); ids one batch ahead of the rows.
// SINCOS (the shipped 'sin_cos' embedding with C % 12 == 0, so that a lane's four channels share their axis): value = sin(2 pi (v_axis * scale_c + phase_c))
// with scale_c = 100 / (dm_c 2 pi), one multiply-add, v_fract and v_sin_f32 per (neighbour, channel); offsets by one reciprocal of the radius.
struct SinCosLane { int axis; float sc[4], ph[4]; };
__device__ inline SinCosLane sincos_lane(int C, int cq)
{
    SinCosLane t;
    const LaneGeo g0 = decode_geo(PE_SIN_COS, C, 4 * cq);
    t.axis = g0.ex;
#pragma unroll
    for (int j = 0; j < 4; j++) { const LaneGeo g = decode_geo(PE_SIN_COS, C, 4 * cq + j); t.sc[j] = (100.f / g.dm) * 0.15915494309189535f; t.ph[j] = g.ph; }
    return t;
}
__device__ __forceinline__ float sin_turns(float t) { return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(t)); }

// MODE 0: any embedding (eval_geo per channel); 1: 'sin_cos' (C % 12 == 0); 2: 'xyz' (C % 12 == 0: a lane's four channels multiply with ONE offset component)
template <int U, int MODE>
__global__ __launch_bounds__(256) void pospool_fwd_v4(unsigned n, int n0, int K, int C4, int c4_0, int L, const float* __restrict__ q, const float* __restrict__ s,
                                                      const int* __restrict__ idx, const float4* __restrict__ f, float radius, int pe, int reduction,
                                                      const int* __restrict__ padding_num, float4* __restrict__ out)
{
    const int pad = (reduction == RED_MEAN) ? *padding_num : 0;
    const int C = 4 * C4;
    const int tpb = 256 / L;
    const int ts = threadIdx.x / L, cl = threadIdx.x - ts * L;
    if (ts >= tpb) return;
    const int cq = c4_0 + cl;
    const LaneGeo g0 = decode_geo(pe, C, 4 * cq), g1 = decode_geo(pe, C, 4 * cq + 1), g2 = decode_geo(pe, C, 4 * cq + 2), g3 = decode_geo(pe, C, 4 * cq + 3);
    SinCosLane sl = {};
    if (MODE == 1) sl = sincos_lane(C, cq);
    if (MODE == 2) sl.axis = g0.ex ? 0 : (g0.ey ? 1 : 2);            // 'xyz': the monomial of this lane's channels is x, y or z
    const float inv_radius = 1.0f / radius;
    const unsigned ntrips = (n + tpb - 1) / tpb;
    const unsigned vend = 8 * cbl_xcd_per(ntrips);
    for (unsigned v = blockIdx.x; v < vend; v += gridDim.x) {
        const unsigned p = cbl_xcd_slot(v, ntrips) * tpb + ts;
        if (p >= n) continue;
        const float qx = q[3 * (size_t)p], qy = q[3 * (size_t)p + 1], qz = q[3 * (size_t)p + 2];
        const int* __restrict__ row = idx + (size_t)p * K;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int cnt = 0;
        int idn[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const int v_ = row[min(u, K - 1)]; idn[u] = (u < K) ? v_ : n0; }
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
                if (!real) id[u] = -1;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (id[u] < 0) continue;                              // shadow neighbours: zero feature row, contribute nothing
                if (MODE == 2) {
                    const float vax = ((sl.axis == 0 ? rx[u] - qx : sl.axis == 1 ? ry[u] - qy : rz[u] - qz)) * inv_radius;
                    acc.x = fmaf(vax, fk[u].x, acc.x); acc.y = fmaf(vax, fk[u].y, acc.y); acc.z = fmaf(vax, fk[u].z, acc.z); acc.w = fmaf(vax, fk[u].w, acc.w);
                } else if (MODE == 1) {
                    const float vax = ((sl.axis == 0 ? rx[u] - qx : sl.axis == 1 ? ry[u] - qy : rz[u] - qz)) * inv_radius;
                    acc.x = fmaf(sin_turns(fmaf(vax, sl.sc[0], sl.ph[0])), fk[u].x, acc.x);
                    acc.y = fmaf(sin_turns(fmaf(vax, sl.sc[1], sl.ph[1])), fk[u].y, acc.y);
                    acc.z = fmaf(sin_turns(fmaf(vax, sl.sc[2], sl.ph[2])), fk[u].z, acc.z);
                    acc.w = fmaf(sin_turns(fmaf(vax, sl.sc[3], sl.ph[3])), fk[u].w, acc.w);
                } else {
                    const float x = (rx[u] - qx) * inv_radius, y = (ry[u] - qy) * inv_radius, z = (rz[u] - qz) * inv_radius;    // :68-70 (one rounded reciprocal, within 1 ulp of the three divisions)
                    acc.x += eval_geo(g0, x, y, z) * fk[u].x;         // :230-235
                    acc.y += eval_geo(g1, x, y, z) * fk[u].y;
                    acc.z += eval_geo(g2, x, y, z) * fk[u].z;
                    acc.w += eval_geo(g3, x, y, z) * fk[u].w;
                }
            }
        }
        const float nn = (reduction == RED_MEAN) ? (float)cnt + 1e-5f : 1.f;      // :238-241
        out[(size_t)p * C4 + cq] = make_float4(acc.x / nn, acc.y / nn, acc.z / nn, acc.w / nn);
    }
}

// ---------------------------------------------------------------- backward (sum / mean) as a gather over the transposed neighbour table
// d out / d f[t, c] = sum over the pairs (p, k) that list t (ascending) of go[p, c] / nn[p] * geo(p, k, c): the reference's tf.gather gradient
// (local_aggregation_operators.py:228-242 under TF autodiff) written, not accumulated — no float atomics, no zero fill, deterministic.
// lane = 4 channels of one target, L lanes per target (a chunk of at most 256 float4 columns), 256 / L targets per trip, trips dealt to the XCDs
// in contiguous eighths of the processing order (as aw_bwd_csr_kernel, local_aggregation.hip).
__global__ __launch_bounds__(256) void pospool_inv_count_kernel(int n, int K, const int* __restrict__ idx, const int* __restrict__ padding_num,
                                                                int reduction, float* __restrict__ inv_nn)
{
    const int pad = (reduction == RED_MEAN) ? *padding_num : 0;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) {
        int cnt = 0;
        if (reduction == RED_MEAN)
            for (int k = 0; k < K; k++) cnt += idx[(size_t)p * K + k] < pad ? 1 : 0;
        inv_nn[p] = (reduction == RED_MEAN) ? 1.0f / ((float)cnt + 1e-5f) : 1.0f;        // :238-241
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void pospool_bwd_csr_kernel(unsigned n0, int C4, int c4_0, int L, CblFastDiv dvK, const float* __restrict__ q,
                                                              const float* __restrict__ s, float radius, int pe, const float* __restrict__ inv_nn,
                                                              const float4* __restrict__ go, const int* __restrict__ order,
                                                              const int* __restrict__ inv_start, const int* __restrict__ inv_src, float4* __restrict__ gf)
{
    const int tpb = 256 / L;
    const int ts = threadIdx.x / L, cl = threadIdx.x - ts * L;
    if (ts >= tpb) return;
    const int cq = c4_0 + cl, C = 4 * C4;
    const LaneGeo g0 = decode_geo(pe, C, 4 * cq), g1 = decode_geo(pe, C, 4 * cq + 1), g2 = decode_geo(pe, C, 4 * cq + 2), g3 = decode_geo(pe, C, 4 * cq + 3);
    SinCosLane sl = {};
    if (MODE == 1) sl = sincos_lane(C, cq);
    if (MODE == 2) sl.axis = g0.ex ? 0 : (g0.ey ? 1 : 2);
    const float inv_radius = 1.0f / radius;
    const unsigned ntrips = (n0 + tpb - 1) / tpb;
    const unsigned vend = 8 * cbl_xcd_per(ntrips);
    for (unsigned v = blockIdx.x; v < vend; v += gridDim.x) {
        const unsigned tr = cbl_xcd_slot(v, ntrips) * tpb + ts;
        if (tr >= n0) continue;
        const int j = order ? order[tr] : (int)tr;
        const int e0 = inv_start[tr], e1 = inv_start[tr + 1];
        const float sx = s[3 * (size_t)j], sy = s[3 * (size_t)j + 1], sz = s[3 * (size_t)j + 2];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        constexpr int U = 4;                                          // pairs in flight per lane
        for (int eb = e0; eb < e1; eb += U) {
            int pi[U]; float4 g[U]; float rx[U], ry[U], rz[U], sc[U];
#pragma unroll
            for (int u = 0; u < U; u++) pi[u] = (int)cbl_fastdiv((unsigned)inv_src[min(eb + u, e1 - 1)], dvK);
#pragma unroll
            for (int u = 0; u < U; u++) {
                g[u] = go[(size_t)pi[u] * C4 + cq];
                rx[u] = q[3 * (size_t)pi[u]]; ry[u] = q[3 * (size_t)pi[u] + 1]; rz[u] = q[3 * (size_t)pi[u] + 2];
                sc[u] = (eb + u < e1) ? inv_nn[pi[u]] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (MODE == 2) {
                    const float vax = ((sl.axis == 0 ? sx - rx[u] : sl.axis == 1 ? sy - ry[u] : sz - rz[u])) * inv_radius * sc[u];
                    acc.x = fmaf(g[u].x, vax, acc.x); acc.y = fmaf(g[u].y, vax, acc.y); acc.z = fmaf(g[u].z, vax, acc.z); acc.w = fmaf(g[u].w, vax, acc.w);
                    continue;
                }
                if (MODE == 1) {
                    const float vax = ((sl.axis == 0 ? sx - rx[u] : sl.axis == 1 ? sy - ry[u] : sz - rz[u])) * inv_radius;
                    acc.x = fmaf(g[u].x * sc[u], sin_turns(fmaf(vax, sl.sc[0], sl.ph[0])), acc.x);
                    acc.y = fmaf(g[u].y * sc[u], sin_turns(fmaf(vax, sl.sc[1], sl.ph[1])), acc.y);
                    acc.z = fmaf(g[u].z * sc[u], sin_turns(fmaf(vax, sl.sc[2], sl.ph[2])), acc.z);
                    acc.w = fmaf(g[u].w * sc[u], sin_turns(fmaf(vax, sl.sc[3], sl.ph[3])), acc.w);
                    continue;
                }
                const float x = (sx - rx[u]) * inv_radius, y = (sy - ry[u]) * inv_radius, z = (sz - rz[u]) * inv_radius;      // :68-70
                acc.x += (g[u].x * sc[u]) * eval_geo(g0, x, y, z);
                acc.y += (g[u].y * sc[u]) * eval_geo(g1, x, y, z);
                acc.z += (g[u].z * sc[u]) * eval_geo(g2, x, y, z);
                acc.w += (g[u].w * sc[u]) * eval_geo(g3, x, y, z);
            }
        }
        gf[(size_t)j * C4 + cq] = acc;
    }
}

inline unsigned pp_grid(int n) { return (unsigned)min((long long)cbl_div_up(n, 4), 256LL * 16); }

int pospool_check(int n, int n0, int K, int C, float radius, int pe, int reduction)
{
    if (n < 0 || n0 < 0 || K <= 0 || K > PP_KMAX || C <= 0 || !(radius > 0.f)) return CBL_ERR_BAD_ARG;
    if (pe < 0 || pe >= PE_COUNT || reduction < 0 || reduction > RED_MAX) return CBL_ERR_BAD_ARG;
    if (pospool_mid(pe, C) == 0) return CBL_ERR_UNSUPPORTED;                      // the reference's reshape (:229) fails for this C
    return CBL_OK;
}

}  // namespace

CBL_EXPORT int cbl_pospool_forward(int n, int n0, int K, int C, const float* query_points, const float* support_points, const int* neighbors_indices,
                                   const float* features, float radius, int position_embedding, int reduction, const int* padding_num,
                                   float* out, void* stream)
{
    const int rc = pospool_check(n, n0, K, C, radius, position_embedding, reduction);
    if (rc) return rc;
    if (n == 0) return CBL_OK;
    if (!query_points || !support_points || !neighbors_indices || !features || !out || (reduction == RED_MEAN && !padding_num)) return CBL_ERR_BAD_ARG;
    const bool vec = reduction != RED_MAX && (C % 4 == 0) && ((((uintptr_t)features | (uintptr_t)out) & 15) == 0);
    if (vec) {
        const int C4 = C / 4, chunks = (C4 + 255) / 256, Lmax = (C4 + chunks - 1) / chunks;
        for (int c4_0 = 0; c4_0 < C4; c4_0 += Lmax) {
            const int L = min(Lmax, C4 - c4_0);
            unsigned g = cbl_round_up8(cbl_div_up(n, 256 / L)); if (g > 8192u) g = 8192u;
#define CBL_PPF(SC_) hipLaunchKernelGGL((pospool_fwd_v4<2, SC_>), dim3(g), dim3(256), 0, cbl_stream(stream), (unsigned)n, n0, K, C4, c4_0, L, query_points, support_points, \
                               neighbors_indices, reinterpret_cast<const float4*>(features), radius, position_embedding, reduction, padding_num, \
                               reinterpret_cast<float4*>(out))
            if (position_embedding == PE_SIN_COS && C % 12 == 0) CBL_PPF(1); else if (position_embedding == PE_XYZ && C % 12 == 0) CBL_PPF(2); else CBL_PPF(0);
#undef CBL_PPF
        }
    }
    else
        hipLaunchKernelGGL(pospool_kernel<false>, dim3(pp_grid(n)), dim3(256), 0, cbl_stream(stream), n, n0, K, C, query_points, support_points,
                           neighbors_indices, features, radius, position_embedding, reduction, padding_num, out, nullptr, nullptr);
    return cbl_status();
}

CBL_EXPORT int cbl_pospool_backward(int n, int n0, int K, int C, const float* query_points, const float* support_points, const int* neighbors_indices,
                                    const float* features, float radius, int position_embedding, int reduction, const int* padding_num,
                                    const float* grad_out, float* grad_features, void* stream)
{
    const int rc = pospool_check(n, n0, K, C, radius, position_embedding, reduction);
    if (rc) return rc;
    if (n == 0) return CBL_OK;
    if (!query_points || !support_points || !neighbors_indices || !features || !grad_out || !grad_features || (reduction == RED_MEAN && !padding_num))
        return CBL_ERR_BAD_ARG;
    hipLaunchKernelGGL(pospool_kernel<true>, dim3(pp_grid(n)), dim3(256), 0, cbl_stream(stream), n, n0, K, C, query_points, support_points,
                       neighbors_indices, features, radius, position_embedding, reduction, padding_num, nullptr, grad_out, grad_features);
    return cbl_status();
}

CBL_EXPORT size_t cbl_pospool_backward_csr_workspace_bytes(int n) { return (((size_t)(n > 0 ? n : 0) + 255) & ~(size_t)255) * sizeof(float); }

CBL_EXPORT int cbl_pospool_backward_csr(int n, int n0, int K, int C, const float* query_points, const float* support_points, const int* neighbors_indices,
                                        float radius, int position_embedding, int reduction, const int* padding_num, const float* grad_out,
                                        const int* order_dst, const int* inv_start, const int* inv_src, float* grad_features,
                                        void* workspace, size_t workspace_bytes, void* stream)
{
    const int rc = pospool_check(n, n0, K, C, radius, position_embedding, reduction);
    if (rc) return rc;
    if (reduction == RED_MAX || C % 4) return CBL_ERR_UNSUPPORTED;                 // 'max' shares a gradient among ties of a row: cbl_pospool_backward
    if (n == 0 || n0 == 0) return CBL_OK;
    if (!query_points || !support_points || !neighbors_indices || !grad_out || !grad_features || !inv_start || !inv_src || !workspace ||
        (reduction == RED_MEAN && !padding_num)) return CBL_ERR_BAD_ARG;
    if (workspace_bytes < cbl_pospool_backward_csr_workspace_bytes(n) || !cbl_host_aligned16(grad_out) || !cbl_host_aligned16(grad_features)) return CBL_ERR_BAD_ARG;
    hipStream_t st = cbl_stream(stream);
    float* inv_nn = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(pospool_inv_count_kernel, dim3(cbl_grid_for(n, 256)), dim3(256), 0, st, n, K, neighbors_indices, padding_num, reduction, inv_nn);
    const int C4 = C / 4;
    const CblFastDiv dv = cbl_fastdiv_make((unsigned)K);
    const int chunks = (C4 + 255) / 256;
    const int Lmax = (C4 + chunks - 1) / chunks;
    for (int c4_0 = 0; c4_0 < C4; c4_0 += Lmax) {
        const int L = min(Lmax, C4 - c4_0);
        unsigned g = cbl_round_up8(cbl_div_up(n0, 256 / L)); if (g > 2048u) g = 2048u;
#define CBL_PPB(SC_) hipLaunchKernelGGL(pospool_bwd_csr_kernel<SC_>, dim3(g), dim3(256), 0, st, (unsigned)n0, C4, c4_0, L, dv, query_points, support_points, radius, \
                           position_embedding, inv_nn, reinterpret_cast<const float4*>(grad_out), order_dst, inv_start, inv_src, \
                           reinterpret_cast<float4*>(grad_features))
        if (position_embedding == PE_SIN_COS && C % 12 == 0) CBL_PPB(1); else if (position_embedding == PE_XYZ && C % 12 == 0) CBL_PPB(2); else CBL_PPB(0);
#undef CBL_PPB
    }
    return cbl_status();
}
