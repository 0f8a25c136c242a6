// K2 (large clouds): furthest point sampling with spatial bucket pruning — the SAME sample sequence as fps.hip / the reference
// (furthestsampling_cuda_kernel, /root/reference/pytorch/lib/pointops/src/sampling/sampling_cuda_kernel.cu:14-129), ties included.
//
// The dense kernel touches every point for every sample (n*m distance updates, 70 ms at 40960 -> 10240).  But a new sample s only
// lowers the running distance t[p] of points closer to it than t[p], i.e. of points within about one current sample spacing.
// So the cloud is cut into buckets of 64 points that are close in space (Morton order), each with a tight bounding box and its
// best entry (max t, smallest reference rank among equals).  Per sample:
//   S1  every lane tests the bucket it owns: L = squared distance from s to the box, computed with the reference's own expression
//       on the box gaps.  Rounding is monotone, so L <= fl(d2(s,p)) for every p in the box with no safety margin, and L >= max t
//       of the bucket proves that nothing in it changes: skip.
//   S2  the owner wave reprocesses each touched bucket: 64 points {x,y,z,t} in one coalesced 1 KiB read from L2, t = min(t, d2),
//       new bucket best by DPP reductions into the owner lane's registers, changed t written back.
//   S3  the block maximum over the bucket bests (same lexicographic key as the dense kernel) is the next sample; it travels with
//       its coordinates through one LDS slot per wave.
// Work drops from n to ~(n/64 box tests + a few hundred point updates) per sample (S-room 40960 -> 10240: 8 touched buckets per
// sample on average); the t values, and therefore the arg-max sequence, are identical by construction.  One barrier and one L2
// round trip per sample.
#include "cbl_common.h"
#include "fps_wave.h"
#include <cstring>
#include <cstdlib>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <math.h>

int cbl_bbox_keys_launch(int b, int n, const float* xyz, const int* offset, unsigned* bbox, hipStream_t st);   // knn_grid.hip

namespace {

constexpr int FB_BUCKET = 64;
constexpr int FB_MAX_BUCKETS = 2048;                                  // n_max <= 131072

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

struct FbWs {
    unsigned* bbox;                                                   // [6*b] order-preserving keys
    unsigned long long* keys_in; unsigned long long* keys_out;        // [n]
    int* vals_in; int* vals_out;                                      // [n]
    float4* sorted;                                                   // [n] {x, y, z, running distance}
    unsigned* rank;                                                   // [n] reference tie rank of the sorted point
    void* cub; size_t cub_bytes;
    size_t bytes;
};

FbWs carve_fb(void* base, int b, int n)
{
    FbWs w;
    char* p = reinterpret_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) { char* r = p ? p + off : nullptr; off += up256(bytes); return r; };
    w.bbox = reinterpret_cast<unsigned*>(take(sizeof(unsigned) * 6 * (size_t)b));
    w.keys_in = reinterpret_cast<unsigned long long*>(take(8 * (size_t)n));
    w.keys_out = reinterpret_cast<unsigned long long*>(take(8 * (size_t)n));
    w.vals_in = reinterpret_cast<int*>(take(4 * (size_t)n));
    w.vals_out = reinterpret_cast<int*>(take(4 * (size_t)n));
    w.sorted = reinterpret_cast<float4*>(take(16 * (size_t)n));
    w.rank = reinterpret_cast<unsigned*>(take(4 * (size_t)n));
    size_t sort_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, w.keys_in, w.keys_out, w.vals_in, w.vals_out, (size_t)(n > 0 ? n : 1));
    w.cub_bytes = sort_bytes;
    w.cub = take(w.cub_bytes + 256);
    w.bytes = off;
    return w;
}

__device__ __forceinline__ float key2f(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__global__ void fb_init_kernel(int b, unsigned* __restrict__ bbox)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 6 * b) bbox[i] = ((i % 6) < 3) ? 0xffffffffu : 0u;
}

__device__ __forceinline__ unsigned spread10(unsigned v)              // 10 bits -> every third bit
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// key = cloud << 32 | 30-bit Morton code of the point in its cloud's bounding cube (1024 cells per axis)
__global__ __launch_bounds__(256) void fb_keys_kernel(int b, int n, const float* __restrict__ xyz, const int* __restrict__ offset,
                                                      const unsigned* __restrict__ bbox, unsigned long long* __restrict__ keys, int* __restrict__ vals)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int c = cbl_cloud_of(i, offset, b);
        const float lx = key2f(bbox[6 * c]), ly = key2f(bbox[6 * c + 1]), lz = key2f(bbox[6 * c + 2]);
        const float ext = fmaxf(fmaxf(key2f(bbox[6 * c + 3]) - lx, key2f(bbox[6 * c + 4]) - ly), key2f(bbox[6 * c + 5]) - lz);
        const float sc = (ext > 0.f) ? 1023.99f / ext : 0.f;
        const unsigned qx = (unsigned)min(1023, max(0, (int)((xyz[3 * i] - lx) * sc)));
        const unsigned qy = (unsigned)min(1023, max(0, (int)((xyz[3 * i + 1] - ly) * sc)));
        const unsigned qz = (unsigned)min(1023, max(0, (int)((xyz[3 * i + 2] - lz) * sc)));
        keys[i] = ((unsigned long long)c << 32) | (spread10(qx) | (spread10(qy) << 1) | (spread10(qz) << 2));
        vals[i] = i;
    }
}

// rank of local point kk under reference block size 2^bits (fps.hip): smaller = preferred among equal distances
__device__ __forceinline__ unsigned fb_rank(int kk, int bits)
{
    const unsigned t = (unsigned)kk & ((1u << bits) - 1u);
    const unsigned j = (unsigned)kk >> bits;
    const unsigned rev = bits ? (__brev(t) >> (32 - bits)) : 0u;
    return (rev << 22) | j;
}
__device__ __forceinline__ int fb_unrank(unsigned rank, int bits)
{
    const unsigned rev = rank >> 22, j = rank & ((1u << 22) - 1u);
    const unsigned t = bits ? (__brev(rev) >> (32 - bits)) : 0u;
    return (int)((j << bits) | t);
}

__global__ __launch_bounds__(256) void fb_gather_kernel(int b, int n, int bits, const float* __restrict__ xyz, const int* __restrict__ offset,
                                                        const float* __restrict__ tmp, const unsigned long long* __restrict__ keys_sorted,
                                                        const int* __restrict__ order, float4* __restrict__ sorted, unsigned* __restrict__ rank)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int src = order[i];
        const int c = (int)(keys_sorted[i] >> 32);
        const int n0 = c ? offset[c - 1] : 0;
        sorted[i] = make_float4(xyz[3 * src], xyz[3 * src + 1], xyz[3 * src + 2], tmp[src]);
        rank[i] = fb_rank(src - n0, bits);
    }
}

__global__ __launch_bounds__(256) void fb_writeback_kernel(int n, const int* __restrict__ order, const float4* __restrict__ sorted, float* __restrict__ tmp)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) tmp[order[i]] = sorted[i].w;   // side effect of :56
}

// ---- DPP reductions: fps_wave.h; the minima over floats are used once per bucket (its box), outside the sample loop
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dppf(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false)); }
__device__ __forceinline__ float wave_min_f(float v)
{
    v = fminf(v, dppf<0xB1, 0xf>(v)); v = fminf(v, dppf<0x4E, 0xf>(v)); v = fminf(v, dppf<0x141, 0xf>(v)); v = fminf(v, dppf<0x140, 0xf>(v));
    v = fminf(v, dppf<0x142, 0xa>(v)); v = fminf(v, dppf<0x143, 0xc>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// coordinates may be negative: the float maximum proper (box corners, once per bucket)
__device__ __forceinline__ float wave_max_any_f(float v)
{
    v = fmaxf(v, dppf<0xB1, 0xf>(v)); v = fmaxf(v, dppf<0x4E, 0xf>(v)); v = fmaxf(v, dppf<0x141, 0xf>(v)); v = fmaxf(v, dppf<0x140, 0xf>(v));
    v = fmaxf(v, dppf<0x142, 0xa>(v)); v = fmaxf(v, dppf<0x143, 0xc>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

struct __attribute__((aligned(16))) FbSlot { float d; unsigned rank; float x, y, z; float pad[3]; };      // 32 B: {d, rank, x, y} and {z, tie flag} are one read each

// One 1024-lane workgroup per cloud.  Bucket g belongs to wave g % 16, lane (g / 16) % 64, register set g / 1024: box, best
// distance / rank / coordinates all live in the owner lane's registers, and the owner WAVE is also the one that reprocesses the
// bucket — so a sample needs no queue, no LDS atomics and ONE barrier: test own buckets -> reprocess the touched ones (ids that
// are neighbours in Morton order sit in different waves) -> wave best -> LDS slot -> barrier -> every wave reduces the 16 slots.
// CERT: also certify which leading samples were unique maxima (the prefix certificate of fps.hip); compiled out otherwise — the bookkeeping costs ~9 % of
// the kernel, and only the head of a sampling chain needs it
template <int R, bool CERT, int W = 16>
__global__ __launch_bounds__(64 * W) void fps_bucket_kernel(int bits, const float* __restrict__ xyz, const int* __restrict__ offset,
                                                          const int* __restrict__ new_offset, float4* __restrict__ sorted,
                                                          const unsigned* __restrict__ rank, int* __restrict__ idx,
                                                          const int* __restrict__ prefix_cert, int* __restrict__ cert_out)
{
    __shared__ FbSlot slots[2][W];
    const int c = blockIdx.x;
    const int n0 = c ? offset[c - 1] : 0, n1 = offset[c];
    const int m0 = c ? new_offset[c - 1] : 0, m1 = new_offset[c];
    if (m1 <= m0 || n1 <= n0) return;
    if (prefix_cert && prefix_cert[c] >= m1 - m0) return;                       // a certified prefix (fps.hip fps_prefix_kernel wrote the samples)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nloc = n1 - n0, NB = (nloc + FB_BUCKET - 1) / FB_BUCKET;

    float lo[R][3], hi[R][3], bm[R], bxr[R], byr[R], bzr[R];
    unsigned brk[R];
    // uniqueness of every arg-max, for the prefix certificate (fps.hip): bt = "this bucket's maximum is attained by more than one of its points"; the same
    // question is carried through the wave's best and the block's best, and the first sample whose maximum was not unique is remembered
    int bt[R];
#pragma unroll
    for (int r = 0; r < R; r++) { bm[r] = -3.f; brk[r] = 0xffffffffu; bxr[r] = byr[r] = bzr[r] = 0.f; bt[r] = 0;
                                  lo[r][0] = lo[r][1] = lo[r][2] = 0.f; hi[r][0] = hi[r][1] = hi[r][2] = 0.f; }

    // the wave recomputes bucket g against sample (sx,sy,sz) (first: no update, also the box); result wave-uniform
    struct Best { float d; unsigned rank; float x, y, z; int tie; };
    auto process = [&](int g, float sx, float sy, float sz, bool first, float* blo, float* bhi) -> Best {
        const int i = n0 + FB_BUCKET * g + lane;
        const bool valid = i < n1;
        const int ic = valid ? i : n1 - 1;
        const float4 p = sorted[ic];
        const unsigned rk = rank[ic];
        float t2 = p.w;
        if (!first) t2 = fminf(cbl_dist2(p.x, p.y, p.z, sx, sy, sz), p.w);      // :54-55
        const float dv = valid ? t2 : -2.f;
        Best o;
        o.d = wave_max_f(dv);
        unsigned long long mk = __ballot(dv == o.d);
        o.tie = CERT ? (__popcll(mk) != 1) : 0;
        if (__popcll(mk) != 1) {                                                 // equal maxima: smallest reference rank wins
            const unsigned wr = wave_min_u(dv == o.d ? rk : 0xffffffffu);
            mk = __ballot(dv == o.d && rk == wr);
        }
        const int lb = __builtin_ctzll(mk);
        o.rank = (unsigned)__builtin_amdgcn_readlane((int)rk, lb);
        o.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p.x), lb));
        o.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p.y), lb));
        o.z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p.z), lb));
        if (first) {                                                             // invalid lanes copy the last valid point: harmless
            blo[0] = wave_min_f(p.x); blo[1] = wave_min_f(p.y); blo[2] = wave_min_f(p.z);
            bhi[0] = wave_max_any_f(p.x); bhi[1] = wave_max_any_f(p.y); bhi[2] = wave_max_any_f(p.z);
        }
        // the write-back goes last and is never waited for (gfx9 counts loads and stores in one counter)
        __builtin_amdgcn_sched_barrier(0);
        if (!first && valid && t2 != p.w) sorted[ic].w = t2;                     // :56
        return o;
    };

    // two buckets at once (update only): same operations as `process`, written side by side so the two chains interleave
    auto process2 = [&](int g1, int g2, float sx, float sy, float sz, Best& o1, Best& o2) {
        const int i1 = n0 + FB_BUCKET * g1 + lane, i2 = n0 + FB_BUCKET * g2 + lane;
        const bool v1 = i1 < n1, v2 = i2 < n1;
        const int c1 = v1 ? i1 : n1 - 1, c2 = v2 ? i2 : n1 - 1;
        const float4 p1 = sorted[c1], p2 = sorted[c2];
        const unsigned rk1 = rank[c1], rk2 = rank[c2];
        const float t1 = fminf(cbl_dist2(p1.x, p1.y, p1.z, sx, sy, sz), p1.w), t2 = fminf(cbl_dist2(p2.x, p2.y, p2.z, sx, sy, sz), p2.w);
        const float d1 = v1 ? t1 : -2.f, d2 = v2 ? t2 : -2.f;
        o1.d = wave_max_f(d1); o2.d = wave_max_f(d2);
        unsigned long long k1 = __ballot(d1 == o1.d), k2 = __ballot(d2 == o2.d);
        o1.tie = CERT ? (__popcll(k1) != 1) : 0; o2.tie = CERT ? (__popcll(k2) != 1) : 0;
        if (__popcll(k1) != 1) { const unsigned wr = wave_min_u(d1 == o1.d ? rk1 : 0xffffffffu); k1 = __ballot(d1 == o1.d && rk1 == wr); }
        if (__popcll(k2) != 1) { const unsigned wr = wave_min_u(d2 == o2.d ? rk2 : 0xffffffffu); k2 = __ballot(d2 == o2.d && rk2 == wr); }
        const int b1 = __builtin_ctzll(k1), b2 = __builtin_ctzll(k2);
        o1.rank = (unsigned)__builtin_amdgcn_readlane((int)rk1, b1); o2.rank = (unsigned)__builtin_amdgcn_readlane((int)rk2, b2);
        o1.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p1.x), b1)); o2.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p2.x), b2));
        o1.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p1.y), b1)); o2.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p2.y), b2));
        o1.z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p1.z), b1)); o2.z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p2.z), b2));
        __builtin_amdgcn_sched_barrier(0);
        if (v1 && t1 != p1.w) sorted[c1].w = t1;
        if (v2 && t2 != p2.w) sorted[c2].w = t2;
    };

#pragma unroll
    for (int r = 0; r < R; r++) {
        for (int l = 0; l < 64; l++) {
            const int g = (l + 64 * r) * W + wave;
            if (g >= NB) break;
            float blo[3], bhi[3];
            const Best o = process(g, 0.f, 0.f, 0.f, true, blo, bhi);
            if (lane == l) { bm[r] = o.d; brk[r] = o.rank; bxr[r] = o.x; byr[r] = o.y; bzr[r] = o.z; bt[r] = o.tie;
#pragma unroll
                             for (int a = 0; a < 3; a++) { lo[r][a] = blo[a]; hi[r][a] = bhi[a]; } }
        }
    }
    float sx = xyz[3 * (size_t)n0], sy = xyz[3 * (size_t)n0 + 1], sz = xyz[3 * (size_t)n0 + 2];   // first sample = first point (:26 / :34)
    int myidx = n0;                                                              // wave 15 collects 64 results per coalesced store (:39)
    // this wave's best over its buckets: per-lane merge of the lane's register sets (cd ..), held by lane `wlane`
    float cd = -3.f, cx = 0.f, cy = 0.f, cz = 0.f; unsigned crk = 0xffffffffu;
    int wlane = 0, fresh = 2;
    float wdv = -4.f;                                                            // the wave's best distance (wave-uniform copy of lane wlane's cd)
    bool wtie = false;
    bool dirty = true;
    int first_tie = 0x7fffffff;                                                  // first sample (counted in the cloud) whose maximum was not unique
    // only the first half of the samples is certified: the next stage of a chain asks for a fraction of them (1/4 in the reference's network), and the
    // bookkeeping sits on the sample's critical path (all of it: 10.9 -> 12.1 ms for 40960 -> 10240)
    const int track_n = (CERT && cert_out) ? (m1 - m0 + 1) >> 1 : 0;

    for (int j = m0 + 1; j < m1; j++) {
        // S1 + S2: own buckets that can change are reprocessed right away
#pragma unroll
        for (int r = 0; r < R; r++) {
            // box gaps; |lo - s| or |s - hi| or 0, squared with the association of cbl_dist2: a lower bound of every fl(d2) in the box
            const float gx = fmaxf(fmaxf(lo[r][0] - sx, sx - hi[r][0]), 0.f);
            const float gy = fmaxf(fmaxf(lo[r][1] - sy, sy - hi[r][1]), 0.f);
            const float gz = fmaxf(fmaxf(lo[r][2] - sz, sz - hi[r][2]), 0.f);
            const float L = (gx * gx + gy * gy) + gz * gz;
            unsigned long long mk = __ballot(L < bm[r]);                        // unowned register sets hold bm = -3: never touched
            // running distances only fall: the wave's best can change only if a touched bucket HELD it (or tied with it: the certificate's uniqueness flag) —
            // the other waves with touched buckets skip the wave reduction and the slot write
            if (mk) dirty |= (__ballot(bm[r] == wdv) & mk) != 0ull;
            while (mk) {
                // up to two touched buckets per trip: their loads and reduction chains overlap
                const int l1 = __builtin_ctzll(mk);
                mk &= mk - 1;
                if (mk) {
                    const int l2 = __builtin_ctzll(mk);
                    mk &= mk - 1;
                    Best o1, o2;
                    process2((l1 + 64 * r) * W + wave, (l2 + 64 * r) * W + wave, sx, sy, sz, o1, o2);
                    if (lane == l1) { bm[r] = o1.d; brk[r] = o1.rank; bxr[r] = o1.x; byr[r] = o1.y; bzr[r] = o1.z; bt[r] = o1.tie; }
                    if (lane == l2) { bm[r] = o2.d; brk[r] = o2.rank; bxr[r] = o2.x; byr[r] = o2.y; bzr[r] = o2.z; bt[r] = o2.tie; }
                } else {
                    const Best o = process((l1 + 64 * r) * W + wave, sx, sy, sz, false, nullptr, nullptr);
                    if (lane == l1) { bm[r] = o.d; brk[r] = o.rank; bxr[r] = o.x; byr[r] = o.y; bzr[r] = o.z; bt[r] = o.tie; }
                }
            }
        }
        if (dirty) {                                                             // wave-uniform
            float d = bm[0]; unsigned rk = brk[0]; float x = bxr[0], y = byr[0], z = bzr[0]; int lt = bt[0];
#pragma unroll
            for (int r = 1; r < R; r++) {
                const bool up = bm[r] > d || (bm[r] == d && brk[r] < rk);
                if (CERT) lt = (bm[r] == d) ? 1 : (up ? bt[r] : lt);             // two of this lane's buckets at the same maximum: not unique
                d = up ? bm[r] : d; rk = up ? brk[r] : rk; x = up ? bxr[r] : x; y = up ? byr[r] : y; z = up ? bzr[r] : z;
            }
            const float wd = wave_max_f(d);
            unsigned long long mk = __ballot(d == wd);
            wtie = CERT && (j - m0 < track_n) && ((__popcll(mk) != 1) || (__ballot(d == wd && lt) != 0ull));
            if (__popcll(mk) != 1) {
                const unsigned wr = wave_min_u(d == wd ? rk : 0xffffffffu);
                mk = __ballot(d == wd && rk == wr);
            }
            // the wave's best stays where it is — in the registers of lane `wlane` — and that lane writes the slot itself: no v_readlane / v_mov detour
            wlane = __builtin_ctzll(mk);
            wdv = wd;
            cd = d; crk = rk; cx = x; cy = y; cz = z;
            dirty = false;
            fresh = 2;
        }
        const int par = j & 1;
        // a wave whose best did not change writes nothing: both parities of its slot hold it after two samples
        if (fresh > 0) {
            fresh--;
            if (lane == wlane) {
                FbSlot* sp = &slots[par][wave];
                *reinterpret_cast<float4*>(sp) = make_float4(cd, __uint_as_float(crk), cx, cy);
                *reinterpret_cast<float2*>(&sp->z) = make_float2(cz, (CERT && wtie) ? 1.f : 0.f);   // the tie flag is part of the wave's state, like its best
            }
        }
        __syncthreads();
        // S3: every wave reduces the 16 slots (each row of 16 lanes holds all of them)
        // lane l reads the whole of slot l & 15 at once; the winner's fields then come out of its lane by v_readlane — no second LDS round trip on the
        // sample's critical path
        const int sl = lane & (W - 1);
        const float4 s4 = *reinterpret_cast<const float4*>(&slots[par][sl]);          // d, rank, x, y
        const float2 s2 = *reinterpret_cast<const float2*>(&slots[par][sl].z);        // z, tie flag
        const float sd = s4.x; const unsigned sr = __float_as_uint(s4.y);
        const float bd = row_max_f(sd);
        constexpr unsigned WM = (1u << W) - 1u;                                    // W < 16: a row holds every slot more than once, the first copy counts
        unsigned mk16 = (unsigned)__ballot(sd == bd) & WM;
        const bool block_tie = __popc(mk16) != 1;
        if (__popc(mk16) != 1) {
            const unsigned br = row_min_u(sd == bd ? sr : 0xffffffffu);
            mk16 = (unsigned)__ballot(sd == bd && sr == br) & WM;
        }
        const int slot = __builtin_ctz(mk16);
        sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s4.z), slot));
        sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s4.w), slot));
        sz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s2.x), slot));
        if (CERT && j - m0 < track_n && first_tie == 0x7fffffff && (block_tie || __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s2.y), slot)) != 0.f)) first_tie = j - m0;
        if (wave == W - 1) {
            const int jj = j - m0;
            const int win = n0 + fb_unrank((unsigned)__builtin_amdgcn_readlane((int)sr, slot), bits);
            if (lane == (jj & 63)) myidx = win;
            if ((jj & 63) == 63 || j == m1 - 1) { if (lane <= (jj & 63)) idx[m0 + (jj & ~63) + lane] = myidx; }
        }
    }
    if (m1 - m0 == 1 && tid == 0) idx[m0] = n0;
    if (CERT && tid == 0 && cert_out) cert_out[c] = min(first_tie, track_n);             // samples 0 .. cert-1 were unique maxima (at most the tracked half)
}

}  // namespace

size_t cbl_fps_bucket_workspace_bytes(int b, int n) { return (b > 0 && n > 0) ? carve_fb(nullptr, b, n).bytes : 0; }

// n = total rows, n_max = largest cloud.  Returns CBL_ERR_UNSUPPORTED when the bucket tables would not fit one workgroup's LDS.
int cbl_fps_bucket_launch(int b, int n, int n_max, int bits, const float* xyz, const int* offset, const int* new_offset, float* tmp, int* idx,
                          void* ws, size_t ws_bytes, hipStream_t st, const int* prefix_cert, int* cert_out)
{
    const int nb_max = (n_max + FB_BUCKET - 1) / FB_BUCKET;
    if (nb_max > FB_MAX_BUCKETS || b > 65535) return CBL_ERR_UNSUPPORTED;
    FbWs w = carve_fb(ws, b, n);
    if (ws_bytes < w.bytes) return CBL_ERR_WORKSPACE;
    const dim3 g(cbl_grid_for(n, 256, 1024)), blk(256);
    hipLaunchKernelGGL(fb_init_kernel, dim3(cbl_div_up(6 * b, 256)), dim3(256), 0, st, b, w.bbox);
    int rc = cbl_bbox_keys_launch(b, n, xyz, offset, w.bbox, st);
    if (rc) return rc;
    hipLaunchKernelGGL(fb_keys_kernel, g, blk, 0, st, b, n, xyz, offset, w.bbox, w.keys_in, w.vals_in);
    size_t cb = w.cub_bytes;
    hipError_t e = rocprim::radix_sort_pairs(w.cub, cb, w.keys_in, w.keys_out, w.vals_in, w.vals_out, (size_t)n, 0u, 48u, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(fb_gather_kernel, g, blk, 0, st, b, n, bits, xyz, offset, tmp, w.keys_out, w.vals_out, w.sorted, w.rank);
    // waves per workgroup: 16 (one bucket per lane up to 1024 buckets); CBL_FPS_WAVES=8 runs two buckets per lane on 8 waves where that covers the cloud
    static const int waves = [] { const char* e = getenv("CBL_FPS_WAVES"); return (e && atoi(e) == 8) ? 8 : 16; }();
#define CBL_FB(R_, W_) do { if (cert_out) hipLaunchKernelGGL((fps_bucket_kernel<R_, true, W_>), dim3(b), dim3(64 * W_), 0, st, bits, xyz, offset, new_offset, w.sorted, w.rank, idx, prefix_cert, cert_out); \
                            else          hipLaunchKernelGGL((fps_bucket_kernel<R_, false, W_>), dim3(b), dim3(64 * W_), 0, st, bits, xyz, offset, new_offset, w.sorted, w.rank, idx, prefix_cert, cert_out); } while (0)
    if (waves == 8 && nb_max <= 1024) { if (nb_max <= 512) CBL_FB(1, 8); else CBL_FB(2, 8); }
    else if (nb_max <= 1024) CBL_FB(1, 16);
    else CBL_FB(2, 16);
#undef CBL_FB
    hipLaunchKernelGGL(fb_writeback_kernel, g, blk, 0, st, n, w.vals_out, w.sorted, tmp);
    return cbl_status();
}
