// K1 (exact): segmented brute-force KNN with the reference's heap semantics, bit for bit.
// Replaces knnquery_cuda_kernel  /root/reference/pytorch/lib/pointops/src/knnquery/knnquery_cuda_kernel.cu:65-111.
//
// MI355X mapping (not the reference's one-thread-per-query): one WAVE per query.  The 64 lanes sweep
// the query's cloud 64 supports at a time (coalesced loads, one distance per lane); a ballot picks the
// lanes whose d2 beats the heap root and they are fed to the heap in ascending lane order — i.e. in the
// reference's visiting order — re-checking against the root, which may have dropped in between.  The
// K-entry max-heap is held ACROSS the lanes of the wave (slot j in lane j, K <= 64) and walked with
// v_readlane / compare-select under wave-uniform control flow: no LDS round trips, no divergence, a few
// cycles per heap access.  For 64 < K <= 1024 the heap sits in LDS (same code shape, broadcast reads).
// Every comparison keeps the reference's strictness (d2 < root; right child only if strictly larger;
// stop only if parent strictly larger), so ties resolve identically — oracle/pointops_oracle.c is the
// CPU statement of the same.  The same kernel replays the queries the grid kernel could not certify
// (worklist mode): m waves spread over the chip instead of one lane scanning a whole cloud.
#include "cbl_common.h"
#include "grid_core.h"
#include <type_traits>

namespace {

constexpr int WAVES_PER_BLOCK = 4;

__device__ __forceinline__ float rl_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ int   rl_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// Heap held across lanes: slot j = (hd, hi) of lane j.
struct LaneHeap {
    float hd; int hi;
    __device__ __forceinline__ void init(int, float d, int i) { hd = d; hi = i; }
    __device__ __forceinline__ float D(int j) const { return rl_f(hd, j); }
    __device__ __forceinline__ int I(int j) const { return rl_i(hi, j); }
    // "writelane" as compare + select: two plain VALU ops, no lane-select hazards to pad
    __device__ __forceinline__ void set(int j, float d, int i) { const bool me = (int)(threadIdx.x & 63) == j; hd = me ? d : hd; hi = me ? i : hi; }
};
// Heap in LDS (one per wave); every lane executes the same accesses (broadcast reads, identical writes).
struct LdsHeap {
    float* d; int* i;
    __device__ __forceinline__ void init(int K, float dv, int iv) { for (int j = threadIdx.x & 63; j < K; j += 64) { d[j] = dv; i[j] = iv; } }
    __device__ __forceinline__ float D(int j) const { return d[j]; }
    __device__ __forceinline__ int I(int j) const { return i[j]; }
    __device__ __forceinline__ void set(int j, float dv, int iv) { d[j] = dv; i[j] = iv; }
};

// Put (d, id) at the root of a max-heap of `len` entries and sift it down ("hole" form of reheap(),
// knnquery_cuda_kernel.cu:21-36: identical final layout, fewer stores).  All operands wave-uniform.
template <class Heap>
__device__ __forceinline__ void heap_replace_root(Heap& h, int len, float d, int id)
{
    int parent = 0;
    for (;;) {
        int kid = 2 * parent + 1;
        if (kid >= len) break;
        float kd = h.D(kid);
        if (kid + 1 < len) {
            const float rd = h.D(kid + 1);
            if (rd > kd) { kd = rd; kid += 1; }          // right child only when strictly larger (:27)
        }
        if (d > kd) break;                                // stop only when strictly larger (:29)
        h.set(parent, kd, h.I(kid));
        parent = kid;
    }
    h.set(parent, d, id);
}

// Ancestor chain of heap slot `lane` as a bit mask over slots (the slot itself up to, excluding, the root).
__device__ __forceinline__ unsigned long long heap_ancestors(int lane)
{
    unsigned long long a = 0;
    for (int c = lane; c >= 1; c = (c - 1) >> 1) a |= 1ull << c;
    return a;
}
// reheap() without the walk: all 64 slots decide at once what they hold after (d, id) replaced the root and sank.
//   * the sink path follows each slot's BIGGER child (right child only when strictly larger, :27).  Slot c is its parent's
//     bigger child  <=>  `isbig` (sibling key through a one-lane DPP shift); it is ON the path  <=>  every slot of its ancestor
//     chain is a bigger child: one ballot, then a mask compare against the precomputed chain;
//   * keys along the path are non-increasing, so the element passes slot c's parent  <=>  !(d > key[c]) (:29), lane-local;
//   * a reached slot takes its bigger child's entry if the element also passes it, else the element itself.
// One LDS-crossbar round trip (the children's entries) and ~20 VALU ops, against a 6-level serial walk.
// The part of the step that depends on the heap alone — each slot's bigger child and whether the slot lies on the sink path —
// and the part that depends on the element.  The replay's feed loop prepares right after every update, so the LDS round trip of the
// preparation runs under the scalar work that finds the next accepted candidate.
struct HeapPrep { float bk; int bi; bool onpath; };
__device__ __forceinline__ HeapPrep heap_prepare(float hd, int hi, int lane, unsigned long long anc, int len)
{
    const int l = 2 * lane + 1, r = l + 1;
    const int al = (l & 63) << 2, ar = (r & 63) << 2;        // ds_bpermute byte addresses (lane-constant: hoisted out of the feed loop)
    const float kl = __int_as_float(__builtin_amdgcn_ds_bpermute(al, __float_as_int(hd)));
    const float kr = __int_as_float(__builtin_amdgcn_ds_bpermute(ar, __float_as_int(hd)));
    const int il = __builtin_amdgcn_ds_bpermute(al, hi), ir = __builtin_amdgcn_ds_bpermute(ar, hi);
    const float up = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(hd), __float_as_int(hd), 0x130, 0xf, 0xf, false));   // slot lane+1
    const float dn = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(hd), __float_as_int(hd), 0x138, 0xf, 0xf, false));   // slot lane-1
    // "is the bigger of its sibling pair" for all slots at once: two vector compares, the rest on 64-bit lane masks on the scalar unit
    // (as bool expressions the lane-constant terms — odd, in heap, has a right sibling — were carried as 0/1 vectors: ~8 extra VALU per insert)
    const unsigned long long m_in = len >= 64 ? ~0ull : ((1ull << len) - 1ull);    // slot < len
    const unsigned long long m_odd = 0xAAAAAAAAAAAAAAAAull;
    const unsigned long long nb = __ballot(up > hd), bp = __ballot(hd > dn);        // right sibling bigger / bigger than the left sibling
    const unsigned long long big = m_in & ((m_odd & ~((m_in >> 1) & nb)) | (~m_odd & bp));
    HeapPrep p;
    p.onpath = (big & anc) == anc;                           // root: empty chain
    const bool right = (r < len) & (kr > kl);                // right child only when strictly larger (:27)
    p.bk = right ? kr : kl;
    p.bi = right ? ir : il;
    return p;
}
__device__ __forceinline__ void heap_apply(float& hd, int& hi, const HeapPrep& p, int lane, int len, float d, int id)
{
    const bool reached = p.onpath & ((lane == 0) | !(d > hd)); // stop only when strictly larger (:29)
    const bool sinks = (2 * lane + 1 < len) & !(d > p.bk);
    const float nd = sinks ? p.bk : d; const int ni = sinks ? p.bi : id;
    hd = reached ? nd : hd; hi = reached ? ni : hi;
}
__device__ __forceinline__ void heap_replace_root_par(float& hd, int& hi, int lane, unsigned long long anc, int len, float d, int id)
{
    const HeapPrep p = heap_prepare(hd, hi, lane, anc, len);
    heap_apply(hd, hi, p, lane, len, d, id);
}

// one query by one wave: the whole scan + heap sort (`heap_mem`: this wave's 2*K words of LDS when the heap does not fit the lanes)
template <bool IN_LANES>
__device__ __forceinline__ void exact_wave_query(int q, int b, int K, const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                 const int* __restrict__ offset, const int* __restrict__ new_offset,
                                                 int* __restrict__ idx, float* __restrict__ dist2, float* heap_mem)
{
    const int lane = threadIdx.x & 63;
    {
    const int c = cbl_cloud_of(q, new_offset, b);
    const int start = (c == 0) ? 0 : offset[c - 1];                  // :75-79
    const int end = offset[c];                                       // :80
    const float qx = new_xyz[3 * q + 0], qy = new_xyz[3 * q + 1], qz = new_xyz[3 * q + 2];

    typename std::conditional<IN_LANES, LaneHeap, LdsHeap>::type h;
    if constexpr (!IN_LANES) {
        h.d = heap_mem;
        h.i = reinterpret_cast<int*>(h.d + K);
    }
    h.init(K, 1e10f, start);                                         // :91-94
    float root = 1e10f;
    const unsigned long long anc = heap_ancestors(lane);
    auto replace_root = [&](int len, float d, int id) {
        if constexpr (IN_LANES) heap_replace_root_par(h.hd, h.hi, lane, anc, len, d, id);
        else heap_replace_root(h, len, d, id);
    };

    // U chunks of 64 supports are loaded ahead of their use: a lone wave (replay mode) would otherwise pay
    // one full memory latency per 64 supports.  Chunks are still consumed strictly in index order.
    constexpr int U = 8;
    for (int base = start; base < end; base += 64 * U) {
        float d2[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            // clamped, UNCONDITIONAL loads: under an exec-masked branch hipcc waits for each chunk's load
            // before issuing the next one; like this all U are in flight together
            const int i = base + 64 * u + lane;
            const int ic = min(i, end - 1);
            const float d = cbl_dist2(qx, qy, qz, xyz[3 * ic + 0], xyz[3 * ic + 1], xyz[3 * ic + 2]);   // (new - x)^2 ..., :99
            d2[u] = (i < end) ? d : INFINITY;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            unsigned long long mask = __ballot(d2[u] < root);        // strict, :100
            while (mask) {                                           // ascending lane = ascending support index
                const int l = __builtin_ctzll(mask);
                mask &= mask - 1;
                const float dl = rl_f(d2[u], l);
                if (dl < root) {                                     // root may have dropped since the ballot
                    replace_root(K, dl, base + 64 * u + l);
                    root = h.D(0);
                }
            }
        }
    }

    // heap_sort(), :39-48
    for (int last = K - 1; last > 0; last--) {
        const float d = h.D(last); const int id = h.I(last);
        h.set(last, h.D(0), h.I(0));
        replace_root(last, d, id);
    }
    int* orow = idx + (size_t)q * K; float* drow = dist2 + (size_t)q * K;
    if constexpr (IN_LANES) {
        if (lane < K) { orow[lane] = h.hi; drow[lane] = h.hd; }
    } else {
        for (int j = lane; j < K; j += 64) { orow[j] = h.i[j]; drow[j] = h.d[j]; }
    }
    }
}

template <bool IN_LANES>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void knn_exact_wave_kernel(
    int b, int m, int K,
    const float* __restrict__ xyz, const float* __restrict__ new_xyz,
    const int* __restrict__ offset, const int* __restrict__ new_offset,
    int* __restrict__ idx, float* __restrict__ dist2,
    const int* __restrict__ worklist, const int* __restrict__ worklist_count)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // work items: all m queries, or (worklist mode) the listed queries, walked grid-stride
    const int n_items = worklist ? *worklist_count : m;
    for (int w = blockIdx.x * WAVES_PER_BLOCK + wave; w < n_items; w += gridDim.x * WAVES_PER_BLOCK) {
        const int q = __builtin_amdgcn_readfirstlane(worklist ? worklist[w] : w);
        exact_wave_query<IN_LANES>(q, b, K, xyz, new_xyz, offset, new_offset, idx, dist2, reinterpret_cast<float*>(smem) + (size_t)wave * 2 * K);
    }
}


// ---------------------------------------------------------------------------------------------------------
// Replay kernel for the handful of queries the grid kernel could not certify: ONE 1024-lane workgroup per query.
// The reference's result depends on the full visiting history, but a support only matters if it beats the heap root at
// its turn, and the root at turn t is the K-th smallest distance among supports [start, t) — non-increasing in t.  So:
//   A. wave 0 feeds the first T0 supports to the heap in order (the dense part of the history);
//      meanwhile waves 1..15 compute B = the K-th smallest distance among those T0 supports (bisection on the float bits): an
//      upper bound of the root for every later turn, hence every later support with d2 >= B is a no-op and can be dropped;
//   B. still under A, waves 1..15 filter the remaining supports against B into per-wave LDS lists, in index order
//      (expected n*K/T0 survivors in total);
//   C. wave 0 feeds the lists to the heap in order, then heap-sorts.
// Identical heap operations as the straight scan, i.e. identical ties; the 40960-support scan no longer sits on one
// wave's memory latency.  The heap insert itself is done without a walk (heap_replace_root_par below).
constexpr int RP_WAVES = 16, RP_T0 = 1024, RP_LIST = 512, RP_MAX_WORK = 1024, RP_GRID = 128;
// Large clouds (n_c > RP_BIG, grid of the search available): the direct filter B stops at support RP_TS.  The root at every later turn is
// at most B* = the K-th smallest distance among the first RP_TS supports — a TIGHT bound (the ball around the query holds ~K n_c / RP_TS
// supports), so the later supports that can matter are enumerated from the few grid cells the ball touches instead of scanning the
// cloud (the scan made a replay O(n_c): 0.3 ms per tied query at a million points), put into index order by a rank sort and fed last.
constexpr int RP_TS = 32768, RP_BIG = 65536, RP_GLIST = 2048, RP_MAX_ROWS = 4096;

// A second job (K2 > 0: another result over the same clouds and queries, with its own worklist) shares the launch: even workgroups take
// job 1, odd ones job 2.  The tied rows of a wide search and of the narrower result derived from it (cbl_knnquery_nested) are both
// known when the search kernel ends, and a replay keeps one or two workgroups busy while 250 CUs idle: side by side, not one after the other.
__global__ __launch_bounds__(64 * RP_WAVES) void knn_replay_kernel(
    int b, int K1, const float* __restrict__ xyz, const float* __restrict__ new_xyz,
    const int* __restrict__ offset, const int* __restrict__ new_offset,
    int* __restrict__ idx1, float* __restrict__ dist2_1,
    const int* __restrict__ worklist1, const int* __restrict__ worklist_count1,
    const CblGrid* __restrict__ grids, const int* __restrict__ cell_start, const float4* __restrict__ sorted,
    int K2, int* __restrict__ idx2, float* __restrict__ dist2_2, const int* __restrict__ worklist2, const int* __restrict__ worklist_count2)
{
    const bool two = K2 > 0, second = two && (blockIdx.x & 1u);
    const int bid = two ? (int)(blockIdx.x >> 1) : (int)blockIdx.x, nbl = two ? (int)(gridDim.x >> 1) : (int)gridDim.x;
    const int K = second ? K2 : K1;
    int* __restrict__ idx = second ? idx2 : idx1;
    float* __restrict__ dist2 = second ? dist2_2 : dist2_1;
    const int* __restrict__ worklist = second ? worklist2 : worklist1;
    const int* __restrict__ worklist_count = second ? worklist_count2 : worklist_count1;
    __shared__ float g_d[2][RP_GLIST];
    __shared__ int g_i[2][RP_GLIST];
    __shared__ int g_count, g_over;
    __shared__ float dA[RP_T0];
    __shared__ float cand_d[RP_WAVES][RP_LIST];
    __shared__ int cand_i[RP_WAVES][RP_LIST];
    __shared__ int cand_n[RP_WAVES];
    __shared__ int overflow_s;

    const int n_work = *worklist_count;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (n_work > RP_MAX_WORK) {
        // long lists (lattices: every query tied): parallelism across entries is plentiful, every wave of the grid takes whole queries
        for (int w = bid * RP_WAVES + wave; w < n_work; w += nbl * RP_WAVES)
            exact_wave_query<true>(__builtin_amdgcn_readfirstlane(worklist[w]), b, K, xyz, new_xyz, offset, new_offset, idx, dist2, nullptr);
        return;
    }
    for (int work = bid; work < n_work; work += nbl) {
    __syncthreads();                                            // previous entry's LDS lists are no longer read
    const int q = worklist[work];
    const int c = cbl_cloud_of(q, new_offset, b);
    const int start = (c == 0) ? 0 : offset[c - 1], end = offset[c];
    const int n_c = end - start;
    const int t0 = min(n_c, RP_T0);
    const bool big = grids != nullptr && n_c > RP_BIG;             // workgroup-uniform
    const int t_star = big ? RP_TS : n_c;                          // the direct filter covers supports [t0, t_star)
    const float qx = new_xyz[3 * q + 0], qy = new_xyz[3 * q + 1], qz = new_xyz[3 * q + 2];

    {   // distances of the first t0 supports, one per lane
        const int i = start + min(tid, max(t0 - 1, 0));
        const float d = (t0 > 0) ? cbl_dist2(qx, qy, qz, xyz[3 * i + 0], xyz[3 * i + 1], xyz[3 * i + 2]) : INFINITY;
        dA[tid] = (tid < t0) ? d : INFINITY;
        if (tid < RP_WAVES) cand_n[tid] = 0;
        if (tid == 0) { overflow_s = 0; g_count = 0; g_over = 0; }
    }
    __syncthreads();

    float hd = 1e10f; int hi = start;                           // heap slot `lane` of wave 0 (:91-94)
    float root = 1e10f;
    const unsigned long long anc = heap_ancestors(lane);
    // feed one chunk: lane l holds candidate l's distance and support index; accepted ones enter the heap in lane order
    HeapPrep prep = heap_prepare(hd, hi, lane, anc, K);         // always describes the current heap while feeding
    auto feed = [&](float d2, int idv) {
        unsigned long long mask = __ballot(d2 < root);          // strict, :100
        while (mask) {
            const int l = __builtin_ctzll(mask);
            mask &= mask - 1;
            const float dl = rl_f(d2, l);
            if (dl < root) {                                    // the root may have dropped since the ballot
                heap_apply(hd, hi, prep, lane, K, dl, rl_i(idv, l));
                root = rl_f(hd, 0);
                prep = heap_prepare(hd, hi, lane, anc, K);      // its LDS round trip overlaps the search for the next candidate
            }
        }
    };

    if (wave == 0) {
        // A (wave 0): the first t0 supports, in order
        for (int base = 0; base < t0; base += 64) feed(dA[base + lane], start + base + lane);
    } else if (n_c > t0) {
        // A' (waves 1..15, each for itself — cheaper than a second barrier): B = K-th smallest of dA, by bisection on the bit
        // patterns (all values >= 0; +inf if fewer than K finite values: nothing can be dropped)
        unsigned v[RP_T0 / 64];
#pragma unroll
        for (int j = 0; j < RP_T0 / 64; j++) v[j] = __float_as_uint(dA[lane + 64 * j]);
        unsigned lo = 0u, hi_b = 0x7f800000u;
        while (lo < hi_b) {
            const unsigned mid = lo + ((hi_b - lo) >> 1);
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < RP_T0 / 64; j++) cnt += __popcll(__ballot(v[j] <= mid));
            if (cnt >= K) hi_b = mid; else lo = mid + 1;
        }
        const float B = __uint_as_float(lo);
        // B (waves 1..15, while wave 0 is still in A): filter a contiguous share of [t0, n_c) against the bound, keeping index order
        const int rem = t_star - t0;
        const int per = ((rem + (RP_WAVES - 1) * 64 - 1) / ((RP_WAVES - 1) * 64)) * 64;
        const int lo_i = start + t0 + (wave - 1) * per, hi_i = min(start + t_star, lo_i + per);
        int filled = 0;
        constexpr int U = 8;
        for (int base = lo_i; base < hi_i; base += 64 * U) {
            float d2[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = base + 64 * u + lane;
                const int ic = min(i, end - 1);
                const float d = cbl_dist2(qx, qy, qz, xyz[3 * ic + 0], xyz[3 * ic + 1], xyz[3 * ic + 2]);
                d2[u] = (i < hi_i) ? d : INFINITY;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const bool keep = d2[u] < B;
                const unsigned long long m = __ballot(keep);
                const int pos = filled + __popcll(m & ((1ull << lane) - 1ull));
                if (keep) {
                    if (pos < RP_LIST) { cand_d[wave][pos] = d2[u]; cand_i[wave][pos] = base + 64 * u + lane; }
                    else overflow_s = 1;
                }
                filled += __popcll(m);
            }
        }
        if (lane == 0) cand_n[wave] = min(filled, RP_LIST);
    }
    __syncthreads();

    // C: the survivors of the direct filter, wave list after wave list = ascending support index (wave 0)
    auto feed_lists = [&]() {
        if (!overflow_s) {
            for (int w = 1; w < RP_WAVES; w++) {
                const int cn = cand_n[w];
                for (int base = 0; base < cn; base += 64) {
                    const int j = min(base + lane, cn - 1);
                    feed((base + lane < cn) ? cand_d[w][j] : INFINITY, cand_i[w][j]);
                }
            }
        } else {
            // a list overflowed (adversarial data, e.g. thousands of supports closer than the first 1024): plain ordered scan
            for (int base = start + t0; base < start + t_star; base += 64) {
                const int i = base + lane;
                const int ic = min(i, end - 1);
                const float d = cbl_dist2(qx, qy, qz, xyz[3 * ic + 0], xyz[3 * ic + 1], xyz[3 * ic + 2]);
                feed((i < start + t_star) ? d : INFINITY, i);
            }
        }
    };
    if (!big) {
        if (wave == 0 && n_c > t0) feed_lists();
    } else {
        if (wave == 0) feed_lists();                                // supports [t0, RP_TS)
        else {
            // B* = K-th smallest distance among dA and the survivors (everything else of the first RP_TS supports is >= B): every wave for
            // itself, values in registers; a subset (more than 2048 survivors) still gives a valid, looser bound
            constexpr int RV = RP_T0 / 64 + 32;
            unsigned v[RV];
#pragma unroll
            for (int j = 0; j < RP_T0 / 64; j++) v[j] = __float_as_uint(dA[lane + 64 * j]);
            {
                int w = 1, base = 0;
#pragma unroll
                for (int j = RP_T0 / 64; j < RV; j++) {
                    while (w < RP_WAVES && base >= cand_n[w]) { w++; base = 0; }      // wave-uniform
                    unsigned val = 0x7f800000u;
                    if (w < RP_WAVES) { if (base + lane < cand_n[w]) val = __float_as_uint(cand_d[w][base + lane]); base += 64; }
                    v[j] = val;
                }
            }
            unsigned lo = 0u, hi_b = 0x7f800000u;
            while (lo < hi_b) {
                const unsigned mid = lo + ((hi_b - lo) >> 1);
                int cnt = 0;
#pragma unroll
                for (int j = 0; j < RV; j++) cnt += __popcll(__ballot(v[j] <= mid));
                if (cnt >= K) hi_b = mid; else lo = mid + 1;
            }
            const float Bs = __uint_as_float(lo);
            // the supports of the ball sqrt(B*) with index >= RP_TS, from the grid rows the ball touches (round-robin over waves 1..15)
            const CblGrid g = grids[c];
            const float uqx = cbl_u(qx, g.ox, g.inv_cs), uqy = cbl_u(qy, g.oy, g.inv_cs), uqz = cbl_u(qz, g.oz, g.inv_cs);
            const int cx = cbl_cell_coord(uqx, g.nx), cy = cbl_cell_coord(uqy, g.ny), cz = cbl_cell_coord(uqz, g.nz);
            const float ru = sqrtf(Bs) * g.inv_cs;
            const int rc = (ru < 1.0e6f) ? (int)ceilf(ru) + 1 : 0x3fffffff;       // one cell of margin for the rounding of the cell coordinates
            const long long side = 2LL * rc + 1;
            if (!(Bs < INFINITY) || side * side > RP_MAX_ROWS) { if (lane == 0) g_over = 1; }
            else {
                const int x0 = max(cx - rc, 0), x1 = min(cx + rc, g.nx - 1);
                const int nrows = (int)(side * side);
                for (int ri = wave - 1; ri < nrows; ri += RP_WAVES - 1) {
                    const int y = cy + ri % (int)side - rc, z = cz + ri / (int)side - rc;
                    if (y < 0 || y >= g.ny || z < 0 || z >= g.nz) continue;
                    const int row = g.cell_base + g.nx * (y + g.ny * z);
                    const int ps = cell_start[row + x0], pe = cell_start[row + x1 + 1];
                    for (int pb = ps; pb < pe; pb += 64) {
                        const int pi = min(pb + lane, pe - 1);
                        const float4 sp = sorted[pi];
                        const float d = cbl_dist2(qx, qy, qz, sp.x, sp.y, sp.z);        // the same floats as xyz[i]: the same distance bits
                        const int si = __float_as_int(sp.w);
                        const bool keep = (pb + lane < pe) && d < Bs && si >= start + RP_TS;
                        const unsigned long long km = __ballot(keep);
                        if (km) {
                            int base = 0;
                            if (lane == 0) base = atomicAdd(&g_count, __popcll(km));
                            base = __builtin_amdgcn_readfirstlane(base);
                            const int pos = base + __popcll(km & ((1ull << lane) - 1ull));
                            if (keep) {
                                if (pos < RP_GLIST) { g_d[0][pos] = d; g_i[0][pos] = si; }
                                else g_over = 1;
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        // index order: rank of every entry among the entries (indices are distinct), all threads
        const int gm = min(g_count, RP_GLIST);
        if (!g_over) {
            for (int e = tid; e < gm; e += 64 * RP_WAVES) {
                const int me = g_i[0][e];
                int rank = 0;
                for (int j = 0; j < gm; j++) rank += (g_i[0][j] < me) ? 1 : 0;
                g_d[1][rank] = g_d[0][e]; g_i[1][rank] = me;
            }
        }
        __syncthreads();
        if (wave == 0) {
            if (!g_over) {
                for (int base = 0; base < gm; base += 64) {
                    const int j = min(base + lane, gm - 1);
                    feed((base + lane < gm) ? g_d[1][j] : INFINITY, g_i[1][j]);
                }
            } else {
                // the ball is too large for the lists (a bound that did not tighten): plain ordered scan of the rest
                for (int base = start + RP_TS; base < end; base += 64) {
                    const int i = base + lane;
                    const int ic = min(i, end - 1);
                    const float d = cbl_dist2(qx, qy, qz, xyz[3 * ic + 0], xyz[3 * ic + 1], xyz[3 * ic + 2]);
                    feed((i < end) ? d : INFINITY, i);
                }
            }
        }
    }
    if (wave != 0) continue;
    // heap_sort(), :39-48
    for (int last = K - 1; last > 0; last--) {
        const float d = rl_f(hd, last); const int id = rl_i(hi, last);
        const float r0 = rl_f(hd, 0); const int i0 = rl_i(hi, 0);
        if (lane == last) { hd = r0; hi = i0; }
        heap_replace_root_par(hd, hi, lane, anc, last, d, id);
    }
    if (lane < K) { idx[(size_t)q * K + lane] = hi; dist2[(size_t)q * K + lane] = hd; }
    }
}

}  // namespace

static int launch_knn_exact(int b, int m, int K, const float* xyz, const float* new_xyz, const int* offset,
                            const int* new_offset, int* idx, float* dist2,
                            const int* worklist, const int* worklist_count, int max_work, hipStream_t st,
                            const CblGrid* grids = nullptr, const int* cell_start = nullptr, const float4* sorted = nullptr)
{
    const int nq = worklist ? max_work : m;
    if (nq <= 0) return CBL_OK;
    if (worklist && K <= 64) {
        // one launch: a 1024-lane workgroup per entry for short lists (the normal case: a handful of tied queries), a wave per entry for long ones
        hipLaunchKernelGGL(knn_replay_kernel, dim3(min(nq, RP_GRID)), dim3(64 * RP_WAVES), 0, st, b, K, xyz, new_xyz, offset, new_offset, idx, dist2, worklist, worklist_count,
                           grids, cell_start, sorted, 0, nullptr, nullptr, nullptr, nullptr);
        return cbl_status();
    }
    const unsigned blocks = worklist ? (unsigned)min((long long)cbl_div_up(nq, WAVES_PER_BLOCK), 2048LL) : cbl_div_up(nq, WAVES_PER_BLOCK);
    const dim3 grid(blocks), block(64 * WAVES_PER_BLOCK);
    if (K <= 64)
        hipLaunchKernelGGL(knn_exact_wave_kernel<true>, grid, block, 0, st,
                           b, m, K, xyz, new_xyz, offset, new_offset, idx, dist2, worklist, worklist_count);
    else
        hipLaunchKernelGGL(knn_exact_wave_kernel<false>, grid, block, (size_t)WAVES_PER_BLOCK * K * 8, st,
                           b, m, K, xyz, new_xyz, offset, new_offset, idx, dist2, worklist, worklist_count);
    return cbl_status();
}

// used by knn_grid.hip for the exact replay of tied queries
// grids / cell_start / sorted: the search grid over `xyz` if the caller still has it (knn_grid.hip's workspace), else null
int cbl_knn_exact_worklist(int b, int m, int K, const float* xyz, const float* new_xyz, const int* offset,
                           const int* new_offset, int* idx, float* dist2,
                           const int* worklist, const int* worklist_count, int max_work, hipStream_t st,
                           const void* grids, const int* cell_start, const void* sorted)
{
    return launch_knn_exact(b, m, K, xyz, new_xyz, offset, new_offset, idx, dist2, worklist, worklist_count, max_work, st,
                            reinterpret_cast<const CblGrid*>(grids), cell_start, reinterpret_cast<const float4*>(sorted));
}

CBL_EXPORT int cbl_knnquery_exact(int b, int n, int m, int nsample, const float* xyz, const float* new_xyz,
                                  const int* offset, const int* new_offset, int* idx, float* dist2, void* stream)
{
    (void)n;
    if (b <= 0 || m < 0 || nsample <= 0 || nsample > CBL_KNN_MAX_NSAMPLE) return CBL_ERR_BAD_ARG;
    if (m == 0) return CBL_OK;
    if (!xyz || !new_xyz || !offset || !new_offset || !idx || !dist2) return CBL_ERR_BAD_ARG;
    return launch_knn_exact(b, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, nullptr, nullptr, 0, cbl_stream(stream));
}

// two replays over the same clouds and queries in one launch (see knn_replay_kernel): K1, K2 <= 64
int cbl_knn_exact_worklist2(int b, int m, const float* xyz, const float* new_xyz, const int* offset, const int* new_offset,
                            int K1, int* idx1, float* dist2_1, const int* worklist1, const int* count1,
                            int K2, int* idx2, float* dist2_2, const int* worklist2, const int* count2,
                            hipStream_t st, const void* grids, const int* cell_start, const void* sorted)
{
    if (m <= 0) return CBL_OK;
    if (K1 > 64 || K2 > 64 || K1 <= 0 || K2 <= 0) return CBL_ERR_BAD_ARG;
    const int per = min(m, RP_GRID);
    hipLaunchKernelGGL(knn_replay_kernel, dim3(2 * per), dim3(64 * RP_WAVES), 0, st, b, K1, xyz, new_xyz, offset, new_offset, idx1, dist2_1, worklist1, count1,
                       reinterpret_cast<const CblGrid*>(grids), cell_start, reinterpret_cast<const float4*>(sorted), K2, idx2, dist2_2, worklist2, count2);
    return cbl_status();
}
