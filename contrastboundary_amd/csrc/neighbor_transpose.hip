// Transposed neighbour table ("CSR by target") and the scatter-add backward passes rewritten as gathers over it.
//
// The reference's backward kernels are scatter-adds with float atomics:
//   grouping_backward_cuda_kernel      /root/reference/pytorch/lib/pointops/src/grouping/grouping_cuda_kernel.cu:16-25
//   (and, through autograd's index_select backward, the neighbour half of the CBL gradient, pytorch/model/heads.py:185-246)
// On MI355X a device-scope float atomic is executed on the memory side of the fabric (the eight XCD L2s are not coherent): measured in
// round 1, 16.9 M lane-atomics = 68 MB of WRITE_SIZE for a 5 MB gradient.  SURVEY.md §7 hard part 6 prescribes the inverse index instead:
// build, once per neighbour table, for every TARGET row the list of (source, column) pairs that point at it; every scatter-add then becomes
// a gather with a segmented sum — no atomics, no zero fill, run-to-run deterministic, and (pairs kept in ascending order) the same summation
// order as the reference's sequential loop on the CPU oracle.
//
// Build (no global atomics per pair, no sort): pairs are binned by TARGET TILE (64 consecutive targets of the processing order) with
// LDS histograms — consecutive sources of the cell order point into a handful of target tiles, so a source tile issues a few dozen global
// atomics for its 64 x nsample pairs — then every target tile orders its bin by (target, pair) in LDS.
//   nt_prep    rank[order[r]] = r, counters zeroed
//   nt_count   per source tile: LDS histogram over target tiles -> reserves a range in every bin it touches (one returning atomic per
//              touched bin) and keeps the (bin, offset) list
//   nt_bin     per source tile: exclusive scan of the bin sizes (redundantly per workgroup, a few hundred ints), then writes
//              (target slot, pair) into its reserved ranges
//   nt_finish  per target tile: counting sort by target in LDS, rank sort by pair inside each target's segment, writes inv_start / inv_src
#include "cbl_common.h"
#include <stdlib.h>

namespace {

constexpr int TT = 64;        // targets per target tile
constexpr int TS = 64;        // sources per source tile
constexpr int NB = 256;       // threads per workgroup
constexpr int NT_MAX_TILES = 16384;    // LDS: 2 x 4 B per target tile in nt_bin
constexpr int NT_SCAN_SPLIT = 2048;    // above this many target tiles the scan of the bin sizes gets a launch of its own (nt_scan_kernel)
constexpr int STAGE_CAP = 12288;       // pair ids a target tile orders in LDS (48 KB); larger bins are ordered through global scratch


struct NtWs {
    int* rank;          // n        (position of a point in `order`; unused without an order)
    int* tile_cursor;   // ntt + 1  (pairs per target tile, accumulated by nt_count)
    int* list_cursor;   // 1
    int* tile_list_off; // nst      (where a source tile's (bin, offset) list starts)
    int* tile_list_n;   // nst
    int* tile_base;     // ntt + 1  (exclusive scan of tile_cursor, written by workgroup 0 of nt_bin)
    int2* lists;        // P        ((bin, offset) records: at most one per pair)
    int2* bins;         // P        ((target slot in its tile, pair))
    int* scratch;       // P        (ordering space for bins beyond STAGE_CAP)
};

static size_t nt_align(size_t v) { return (v + 255) & ~(size_t)255; }
static int nt_tiles(int n) { return (n + TT - 1) / TT; }
static int nt_src_tiles(int m) { return (m + TS - 1) / TS; }

static size_t nt_carve(NtWs& w, char* base, int m, int n, int nsample)
{
    const size_t P = (size_t)m * nsample;
    const int ntt = nt_tiles(n), nst = nt_src_tiles(m);
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += nt_align(bytes); return p; };
    w.rank = (int*)take(sizeof(int) * (size_t)(n > 0 ? n : 1));
    w.tile_cursor = (int*)take(sizeof(int) * (size_t)(ntt + 2));       // + list_cursor right behind it: one zero fill
    w.list_cursor = w.tile_cursor ? w.tile_cursor + ntt + 1 : nullptr;
    w.tile_list_off = (int*)take(sizeof(int) * (size_t)(nst + 1));
    w.tile_list_n = (int*)take(sizeof(int) * (size_t)(nst + 1));
    w.tile_base = (int*)take(sizeof(int) * (size_t)(ntt + 2));
    w.lists = (int2*)take(sizeof(int2) * (P ? P : 1));
    w.bins = (int2*)take(sizeof(int2) * (P ? P : 1));
    w.scratch = (int*)take(sizeof(int) * (P ? P : 1));
    return off;
}

__global__ __launch_bounds__(NB) void nt_prep_kernel(int n, int nzero, const int* __restrict__ order, int* __restrict__ rank, int* __restrict__ zero,
                                                     int* __restrict__ zero2)
{
    const int i = blockIdx.x * NB + threadIdx.x;
    if (order && i < n) rank[order[i]] = i;
    if (i < nzero) { zero[i] = 0; if (zero2) zero2[i] = 0; }          // zero2: the second table of a pair build (cbl_neighbor_transpose_pair)
}

// A source tile's pairs are walked UN pairs per thread at a time — UN chosen so that ONE batch covers the tile (64 x nsample pairs over 256
// threads) — with the three dependent accesses of a pair (its neighbour id, that neighbour's rank, the LDS counter) issued level by
// level: the kernels are chains of L2 round trips, not bandwidth, and every batch is two of them.
template <int UN> struct NtPairs { unsigned p[UN]; int r[UN]; };    // flat pair index and target rank (-1: padding / outside the tile)

template <int UN>
__device__ __forceinline__ void nt_load_pairs(NtPairs<UN>& q, unsigned base, unsigned total, int n, int ns, CblFastDiv dv, const int* __restrict__ src_ids,
                                              const int* __restrict__ idx, const int* __restrict__ rank)
{
    int t[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
        const unsigned e = base + (unsigned)u * NB + threadIdx.x;
        const bool in = e < total;
        const unsigned sl = cbl_fastdiv(in ? e : 0u, dv), col = (in ? e : 0u) - sl * (unsigned)ns;
        q.p[u] = (unsigned)src_ids[sl] * (unsigned)ns + col;
        t[u] = in ? idx[q.p[u]] : -1;
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
        const bool ok = (unsigned)t[u] < (unsigned)n;                // shadow / padding neighbours take no part
        q.r[u] = ok ? (rank ? rank[t[u]] : t[u]) : -1;
    }
}

// one table's share of a pair build: what the kernels below take as separate arguments.  Both tables of a pair have the same sources, targets and orders
// (two neighbour tables of ONE geometry: the block's K = 8 / 16 and the CBL head's K = 36), so rank, order_src and the tile counts are shared.
struct NtJob {
    int ns; CblFastDiv dv; const int* idx;
    int *tile_cursor, *list_cursor, *tile_list_off, *tile_list_n, *tile_base; int2 *lists, *bins; int* scratch;
    int *inv_start, *inv_src;
};

template <int UN>
__device__ __forceinline__ void nt_count_body(int* lds, int m, int n, int ns, int ntt, CblFastDiv dv, const int* __restrict__ idx, const int* __restrict__ order_src,
                                              const int* __restrict__ rank, int* __restrict__ tile_cursor, int* __restrict__ list_cursor,
                                              int* __restrict__ tile_list_off, int* __restrict__ tile_list_n, int2* __restrict__ lists)
{
    int* hist = lds;                                                 // ntt
    int2* mine = reinterpret_cast<int2*>(lds + ((ntt + 1) & ~1));    // up to min(ntt, TS * ns) records (8-byte aligned)
    __shared__ int nmine, gbase, src_ids[TS];
    const int st = blockIdx.x;
    const int nsrc = min(TS, m - st * TS);
    if ((int)threadIdx.x < TS) src_ids[threadIdx.x] = (int)threadIdx.x < nsrc ? (order_src ? order_src[st * TS + threadIdx.x] : st * TS + (int)threadIdx.x) : 0;
    for (int e = threadIdx.x; e < ntt; e += NB) hist[e] = 0;
    if (threadIdx.x == 0) nmine = 0;
    __syncthreads();
    const unsigned total = (unsigned)nsrc * (unsigned)ns;
    for (unsigned base = 0; base < total; base += UN * NB) {
        NtPairs<UN> q;
        nt_load_pairs<UN>(q, base, total, n, ns, dv, src_ids, idx, rank);
#pragma unroll
        for (int u = 0; u < UN; u++) if (q.r[u] >= 0) atomicAdd(&hist[q.r[u] / TT], 1);
    }
    __syncthreads();
    for (int tt = threadIdx.x; tt < ntt; tt += NB) {
        const int c = hist[tt];
        if (c) {
            const int off = atomicAdd(&tile_cursor[tt], c);         // this tile's range inside bin tt (relative to the bin's start)
            mine[atomicAdd(&nmine, 1)] = make_int2(tt, off);
        }
    }
    __syncthreads();
    // a source tile's records live at its own offset (a tile touches at most as many bins as it has pairs): no cursor all tiles would queue on
    if (threadIdx.x == 0) { gbase = st * TS * ns; tile_list_off[st] = gbase; tile_list_n[st] = nmine; }
    __syncthreads();
    for (int e = threadIdx.x; e < nmine; e += NB) lists[gbase + e] = mine[e];
}
template <int UN>
__global__ __launch_bounds__(NB) void nt_count_kernel(int m, int n, int ns, int ntt, CblFastDiv dv, const int* __restrict__ idx, const int* __restrict__ order_src,
                                                      const int* __restrict__ rank, int* __restrict__ tile_cursor, int* __restrict__ list_cursor,
                                                      int* __restrict__ tile_list_off, int* __restrict__ tile_list_n, int2* __restrict__ lists)
{
    extern __shared__ int lds[];
    nt_count_body<UN>(lds, m, n, ns, ntt, dv, idx, order_src, rank, tile_cursor, list_cursor, tile_list_off, tile_list_n, lists);
}
// the two tables of a pair in ONE launch: blockIdx.y picks the table.  The build is a chain of L2 round trips on a few workgroups per CU; side by side the two
// chains cover each other, and a step has four launches less.
template <int UNA, int UNB>
__global__ __launch_bounds__(NB) void nt_count_pair_kernel(int m, int n, int ntt, const int* __restrict__ order_src, const int* __restrict__ rank, NtJob a, NtJob b)
{
    extern __shared__ int lds[];
    if (blockIdx.y == 0) nt_count_body<UNA>(lds, m, n, a.ns, ntt, a.dv, a.idx, order_src, rank, a.tile_cursor, a.list_cursor, a.tile_list_off, a.tile_list_n, a.lists);
    else                 nt_count_body<UNB>(lds, m, n, b.ns, ntt, b.dv, b.idx, order_src, rank, b.tile_cursor, b.list_cursor, b.tile_list_off, b.tile_list_n, b.lists);
}

// exclusive scan of the bin sizes by ONE workgroup (ntt <= NT_MAX_TILES = 16 per lane of 1024): only launched for large tables, where every
// source-tile workgroup scanning all bin sizes itself (nt_bin_kernel<UN, false>) is a term quadratic in the number of tiles — nothing at 640
// tiles, 3 of 10 ms of a million-point step
__global__ __launch_bounds__(1024) void nt_scan_kernel(int ntt, const int* __restrict__ tile_cursor, int* __restrict__ tile_base)
{
    __shared__ int wave_tot[16];
    constexpr int PER = NT_MAX_TILES / 1024;
    const int c0 = threadIdx.x * PER;
    int v[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) { v[k] = (c0 + k < ntt) ? tile_cursor[c0 + k] : 0; sum += v[k]; }
    int incl = sum;
    for (int k = 1; k < 64; k <<= 1) { const int o = __shfl_up(incl, k); if ((int)(threadIdx.x & 63) >= k) incl += o; }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) run += wave_tot[w];
#pragma unroll
    for (int k = 0; k < PER; k++) { if (c0 + k < ntt) tile_base[c0 + k] = run; run += v[k]; }
    if (threadIdx.x == 1023) tile_base[ntt] = run;                   // (its chunk ends at or past ntt: the total number of pairs)
}

template <int UN, bool SCANNED>
__device__ __forceinline__ void nt_bin_body(int* lds, int m, int n, int ns, int ntt, CblFastDiv dv, const int* __restrict__ idx, const int* __restrict__ order_src,
                                            const int* __restrict__ rank, const int* __restrict__ tile_cursor, const int* __restrict__ tile_list_off,
                                            const int* __restrict__ tile_list_n, const int2* __restrict__ lists, int* __restrict__ tile_base,
                                            int2* __restrict__ bins)
{
    int* where = lds;                                                // ntt: size of bin tt -> its start -> (touched bins) start of this tile's range in it
    int* hist = lds + ntt;                                           // ntt: running position inside the range
    __shared__ int wave_tot[NB / 64], src_ids[TS];
    const int st = blockIdx.x;
    const int nsrc = min(TS, m - st * TS);
    if ((int)threadIdx.x < TS) src_ids[threadIdx.x] = (int)threadIdx.x < nsrc ? (order_src ? order_src[st * TS + threadIdx.x] : st * TS + (int)threadIdx.x) : 0;
    if (SCANNED) {                                                   // tile_base holds the scan (nt_scan_kernel): only the bins this tile touches are set up
        const int nl0 = tile_list_n[st], lo0 = tile_list_off[st];
        for (int e = threadIdx.x; e < nl0; e += NB) { const int2 r = lists[lo0 + e]; where[r.x] = tile_base[r.x] + r.y; hist[r.x] = 0; }
        __syncthreads();
    } else {
    for (int e = threadIdx.x; e < ntt; e += NB) { where[e] = tile_cursor[e]; hist[e] = 0; }      // coalesced; the scan below runs out of LDS
    __syncthreads();
    // exclusive scan of the bin sizes: thread t owns a contiguous chunk
    const int per = (ntt + NB - 1) / NB;
    const int c0 = threadIdx.x * per, c1 = min(ntt, c0 + per);
    int sum = 0;
    for (int e = c0; e < c1; e++) sum += where[e];
    int incl = sum;
    for (int k = 1; k < 64; k <<= 1) { const int v = __shfl_up(incl, k); if ((threadIdx.x & 63) >= k) incl += v; }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) run += wave_tot[w];
    for (int e = c0; e < c1; e++) { const int c = where[e]; where[e] = run; if (st == 0) tile_base[e] = run; run += c; }
    if (st == 0 && threadIdx.x == NB - 1) tile_base[ntt] = run;      // the last thread's chunk ends at ntt (possibly empty): total number of pairs
    __syncthreads();
    const int nl = tile_list_n[st], lo = tile_list_off[st];
    for (int e = threadIdx.x; e < nl; e += NB) { const int2 r = lists[lo + e]; where[r.x] += r.y; }
    __syncthreads();
    }
    const unsigned total = (unsigned)nsrc * (unsigned)ns;
    for (unsigned base = 0; base < total; base += UN * NB) {
        NtPairs<UN> q;
        nt_load_pairs<UN>(q, base, total, n, ns, dv, src_ids, idx, rank);
#pragma unroll
        for (int u = 0; u < UN; u++)
            if (q.r[u] >= 0) { const int tt = q.r[u] / TT; bins[where[tt] + atomicAdd(&hist[tt], 1)] = make_int2(q.r[u] & (TT - 1), (int)q.p[u]); }
    }
}
template <int UN, bool SCANNED>
__global__ __launch_bounds__(NB) void nt_bin_kernel(int m, int n, int ns, int ntt, CblFastDiv dv, const int* __restrict__ idx, const int* __restrict__ order_src,
                                                    const int* __restrict__ rank, const int* __restrict__ tile_cursor, const int* __restrict__ tile_list_off,
                                                    const int* __restrict__ tile_list_n, const int2* __restrict__ lists, int* __restrict__ tile_base,
                                                    int2* __restrict__ bins)
{
    extern __shared__ int lds[];
    nt_bin_body<UN, SCANNED>(lds, m, n, ns, ntt, dv, idx, order_src, rank, tile_cursor, tile_list_off, tile_list_n, lists, tile_base, bins);
}
template <int UNA, int UNB>
__global__ __launch_bounds__(NB) void nt_bin_pair_kernel(int m, int n, int ntt, const int* __restrict__ order_src, const int* __restrict__ rank, NtJob a, NtJob b)
{
    extern __shared__ int lds[];
    if (blockIdx.y == 0) nt_bin_body<UNA, false>(lds, m, n, a.ns, ntt, a.dv, a.idx, order_src, rank, a.tile_cursor, a.tile_list_off, a.tile_list_n, a.lists, a.tile_base, a.bins);
    else                 nt_bin_body<UNB, false>(lds, m, n, b.ns, ntt, b.dv, b.idx, order_src, rank, b.tile_cursor, b.tile_list_off, b.tile_list_n, b.lists, b.tile_base, b.bins);
}

// counting sort of a target tile's bin by target, then a rank sort by pair inside every target's segment (pair ids are distinct):
// ascending pairs = the reference loop's summation order.  `stage` is LDS (bins up to STAGE_CAP pairs) or global scratch.  A thread keeps
// its entries in registers between the counting pass and the placing pass when the bin is small enough (EPT per thread), else it reloads.
constexpr int EPT = 16;

// ascending bitonic sort of one int per lane over the 64 lanes ("flip" form: every merge starts with a mirror exchange), exchanges on the
// VALU's data-parallel primitives / ds_swizzle — the network of knn_grid.hip's wave_sort64, keys only
template <int CTRL> __device__ __forceinline__ int nt_dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ int nt_cx(int v, int partner, bool lower) { return lower ? min(v, partner) : max(v, partner); }
__device__ __forceinline__ int nt_wave_sort64(int v, int lane)
{
    const bool l1 = !(lane & 1), l2 = !(lane & 2), l4 = !(lane & 4), l8 = !(lane & 8), l16 = !(lane & 16), l32 = !(lane & 32);
#define X1(v) nt_dpp<0xB1>(v)
#define X2(v) nt_dpp<0x4E>(v)
#define QM(v) nt_dpp<0x1B>(v)
#define HM(v) nt_dpp<0x141>(v)
#define RM(v) nt_dpp<0x140>(v)
#define R8(v) nt_dpp<0x128>(v)
#define S4(v) __builtin_amdgcn_ds_swizzle(v, 0x101f)
#define S16(v) __builtin_amdgcn_ds_swizzle(v, 0x401f)
#define M32(v) __builtin_amdgcn_ds_swizzle(v, 0x7c1f)
#define M64(v) __builtin_amdgcn_ds_bpermute((63 - lane) << 2, v)
    v = nt_cx(v, X1(v), l1);
    v = nt_cx(v, QM(v), l2);  v = nt_cx(v, X1(v), l1);
    v = nt_cx(v, HM(v), l4);  v = nt_cx(v, X2(v), l2);  v = nt_cx(v, X1(v), l1);
    v = nt_cx(v, RM(v), l8);  v = nt_cx(v, S4(v), l4);  v = nt_cx(v, X2(v), l2);  v = nt_cx(v, X1(v), l1);
    v = nt_cx(v, M32(v), l16); v = nt_cx(v, R8(v), l8); v = nt_cx(v, S4(v), l4);  v = nt_cx(v, X2(v), l2); v = nt_cx(v, X1(v), l1);
    v = nt_cx(v, M64(v), l32); v = nt_cx(v, S16(v), l16); v = nt_cx(v, R8(v), l8); v = nt_cx(v, S4(v), l4); v = nt_cx(v, X2(v), l2); v = nt_cx(v, X1(v), l1);
#undef X1
#undef X2
#undef QM
#undef HM
#undef RM
#undef R8
#undef S4
#undef S16
#undef M32
#undef M64
    return v;
}

// every target's segment into ascending pair order (pair ids are distinct): one wave per target, the segment in registers (up to 64
// pairs: a 21-stage exchange network; the rank sort through LDS it replaces cost 13 of the kernel's 19 us), longer segments by rank
__device__ __forceinline__ void nt_rank_sort(const int* __restrict__ stage, const int* lstart, int* __restrict__ out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // four segments at a time per wave: the network is 21 DEPENDENT exchanges (a third of them through the LDS crossbar), four independent
    // sorts interleaved cover each other's latency
    for (int base = wave * (TT / (NB / 64)); base < (wave + 1) * (TT / (NB / 64)); base += 4) {
        int v[4], s0v[4], Lv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            s0v[u] = lstart[base + u]; Lv[u] = lstart[base + u + 1] - s0v[u];
            v[u] = (lane < Lv[u] && Lv[u] <= 64) ? stage[s0v[u] + lane] : 0x7fffffff;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = nt_wave_sort64(v[u], lane);
#pragma unroll
        for (int u = 0; u < 4; u++) if (Lv[u] <= 64 && lane < Lv[u]) out[s0v[u] + lane] = v[u];
    }
    for (int slot = wave; slot < TT; slot += NB / 64) {              // segments longer than a wave: by rank
        const int s0 = lstart[slot], L = lstart[slot + 1] - s0;
        if (L <= 64) continue;
        for (int c = 0; c < L; c += 64) {
            const int e = c + lane;
            const int mine = e < L ? stage[s0 + e] : 0x7fffffff;
            int rnk = 0, o = 0;
            for (; o + 8 <= L; o += 8) {                             // same address for every lane: LDS broadcast reads, eight in flight
                int v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = stage[s0 + o + u];
#pragma unroll
                for (int u = 0; u < 8; u++) rnk += (v[u] < mine) ? 1 : 0;
            }
            for (; o < L; o++) rnk += (stage[s0 + o] < mine) ? 1 : 0;
            if (e < L) out[s0 + rnk] = mine;
        }
    }
}

__device__ __forceinline__ void nt_finish_body(int n, int ntt, const int* __restrict__ tile_base, const int2* __restrict__ bins, int* __restrict__ scratch,
                                               int* __restrict__ inv_start, int* __restrict__ inv_src)
{
    __shared__ int cnt[TT], lstart[TT + 1];
    __shared__ int stage_lds[STAGE_CAP];
    const int tt = blockIdx.x;
    const int b0 = tile_base[tt], E = tile_base[tt + 1] - b0;
    const bool in_regs = E <= EPT * NB;                              // block-uniform
    if (threadIdx.x < TT) cnt[threadIdx.x] = 0;
    int2 r[EPT];
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < EPT; u++) { const int e = u * NB + (int)threadIdx.x; r[u] = e < E ? bins[b0 + e] : make_int2(-1, 0); }   // one batch of loads
    }
    __syncthreads();
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < EPT; u++) if (r[u].x >= 0) atomicAdd(&cnt[r[u].x], 1);
    } else {
        for (int e = threadIdx.x; e < E; e += NB) atomicAdd(&cnt[bins[b0 + e].x], 1);
    }
    __syncthreads();
    if (threadIdx.x < 64) {                                          // TT == 64: one wave scans the counts
        const int c = cnt[threadIdx.x];
        int incl = c;
        for (int k = 1; k < 64; k <<= 1) { const int v = __shfl_up(incl, k); if ((int)threadIdx.x >= k) incl += v; }
        lstart[threadIdx.x] = incl - c;
        if (threadIdx.x == 63) lstart[TT] = incl;
        const int row = tt * TT + threadIdx.x;
        if (row < n) inv_start[row] = b0 + incl - c;
        if (tt == ntt - 1 && threadIdx.x == 63) inv_start[n] = b0 + incl;
        cnt[threadIdx.x] = 0;
    }
    __syncthreads();
    if (E <= STAGE_CAP) {                                            // (EPT * NB <= STAGE_CAP: a register-resident bin is always staged in LDS)
        if (in_regs) {
#pragma unroll
            for (int u = 0; u < EPT; u++) if (r[u].x >= 0) stage_lds[lstart[r[u].x] + atomicAdd(&cnt[r[u].x], 1)] = r[u].y;
        } else {
            for (int e = threadIdx.x; e < E; e += NB) { const int2 v = bins[b0 + e]; stage_lds[lstart[v.x] + atomicAdd(&cnt[v.x], 1)] = v.y; }
        }
        __syncthreads();
        nt_rank_sort(stage_lds, lstart, inv_src + b0);
    } else {
        int* stage = scratch + b0;
        for (int e = threadIdx.x; e < E; e += NB) { const int2 v = bins[b0 + e]; stage[lstart[v.x] + atomicAdd(&cnt[v.x], 1)] = v.y; }
        __threadfence_block();
        __syncthreads();
        nt_rank_sort(stage, lstart, inv_src + b0);
    }
}
__global__ __launch_bounds__(NB) void nt_finish_kernel(int n, int ntt, const int* __restrict__ tile_base, const int2* __restrict__ bins, int* __restrict__ scratch,
                                                       int* __restrict__ inv_start, int* __restrict__ inv_src)
{
    nt_finish_body(n, ntt, tile_base, bins, scratch, inv_start, inv_src);
}
__global__ __launch_bounds__(NB) void nt_finish_pair_kernel(int n, int ntt, NtJob a, NtJob b)
{
    const bool fa = blockIdx.y == 0;                                  // (one call of the body: its 48 KB staging tile exists once)
    nt_finish_body(n, ntt, fa ? a.tile_base : b.tile_base, fa ? a.bins : b.bins, fa ? a.scratch : b.scratch, fa ? a.inv_start : b.inv_start, fa ? a.inv_src : b.inv_src);
}

// ---------------------------------------------------------------- K4 as a gather
// grad_in[t, :] = sum over the pairs p of target t (ascending) of grad_out[p, :]          grouping_cuda_kernel.cu:16-25
// lane = 4 channels of one target; LG = c / 4 lanes per target, 64 / LG targets per wave; sequence slots dealt to the XCDs in contiguous eighths
template <int LG>
__global__ __launch_bounds__(NB) void grouping_bwd_csr_kernel(unsigned n, const float4* __restrict__ go, const int* __restrict__ order,
                                                              const int* __restrict__ inv_start, const int* __restrict__ inv_src, float4* __restrict__ gi)
{
    constexpr unsigned TPB = NB / LG;                                // targets per workgroup
    const unsigned nwg = (n + TPB - 1) / TPB;
    const unsigned g = threadIdx.x / LG, ch = threadIdx.x % LG;
    for (unsigned v = blockIdx.x; v < 8 * cbl_xcd_per(nwg); v += gridDim.x) {
        const unsigned r = cbl_xcd_slot(v, nwg) * TPB + g;
        if (r >= n) continue;
        const int s0 = inv_start[r], s1 = inv_start[r + 1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int e = s0;
        for (; e + 4 <= s1; e += 4) {                                // four independent row loads in flight
            const int p0 = inv_src[e], p1 = inv_src[e + 1], p2 = inv_src[e + 2], p3 = inv_src[e + 3];
            const float4 a = go[(size_t)p0 * LG + ch], b = go[(size_t)p1 * LG + ch], c = go[(size_t)p2 * LG + ch], d = go[(size_t)p3 * LG + ch];
            acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
            acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
            acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += c.w;
            acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
        }
        for (; e < s1; e++) { const float4 a = go[(size_t)inv_src[e] * LG + ch]; acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }
        const unsigned t = order ? (unsigned)order[r] : r;
        gi[(size_t)t * LG + ch] = acc;
    }
}

// any channel count: lane = one channel of one target
__global__ __launch_bounds__(NB) void grouping_bwd_csr_scalar_kernel(unsigned n, int c, const float* __restrict__ go, const int* __restrict__ order,
                                                                     const int* __restrict__ inv_start, const int* __restrict__ inv_src, float* __restrict__ gi)
{
    const unsigned long long total = (unsigned long long)n * (unsigned)c;
    for (unsigned long long e = (unsigned long long)blockIdx.x * NB + threadIdx.x; e < total; e += (unsigned long long)gridDim.x * NB) {
        const unsigned r = (unsigned)(e / (unsigned)c), ch = (unsigned)(e - (unsigned long long)r * (unsigned)c);
        float acc = 0.f;
        for (int k = inv_start[r]; k < inv_start[r + 1]; k++) acc += go[(size_t)inv_src[k] * c + ch];
        gi[(size_t)(order ? order[r] : (int)r) * c + ch] = acc;
    }
}

#include "k4_rows_pipe.h"
static_assert(K4_ROWS_BLOCK == NB, "one block size for the table kernels");

// The other scatter-adds of the path in the same form.  K6 (interpolation_cuda_kernel.cu:20-33) and the grad_input part of K10
// (aggregation_cuda_kernel.cu:22-39) add  rows[source] * weight[pair]  into the pair's target, K8 (subtraction_cuda_kernel.cu:18-30) adds
// -rows[pair]: one wave per target, lane = channel, the target's pairs in ascending order (= the reference loop run sequentially: bit-exact
// against the CPU oracle), eight rows in flight.
template <bool PAIR_ROWS, bool WEIGHT>
__global__ __launch_bounds__(NB) void scatter_as_gather_kernel(unsigned n, int c, CblFastDiv dv, int wc, float sign, const float* __restrict__ rows,
                                                               const float* __restrict__ w, const int* __restrict__ order,
                                                               const int* __restrict__ inv_start, const int* __restrict__ inv_src, float* __restrict__ gi)
{
    const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nwg = (n + 3) >> 2;
    for (unsigned v = blockIdx.x; v < 8 * cbl_xcd_per(nwg); v += gridDim.x) {
        const unsigned r = cbl_xcd_slot(v, nwg) * 4 + wave;
        if (r >= n) continue;
        const int s0 = inv_start[r], s1 = inv_start[r + 1];
        const unsigned t = order ? (unsigned)order[r] : r;
        for (unsigned ch = lane; ch < (unsigned)c; ch += 64) {
            const unsigned wch = WEIGHT ? ch % (unsigned)wc : 0u;
            auto term = [&](int e) -> float {
                const unsigned p = (unsigned)inv_src[e];
                const float x = rows[(size_t)(PAIR_ROWS ? p : cbl_fastdiv(p, dv)) * c + ch];
                return WEIGHT ? x * w[(size_t)p * wc + wch] : x;
            };
            float acc = 0.f;
            int e = s0;
            for (; e + 8 <= s1; e += 8) {
                float x[8];
#pragma unroll
                for (int u = 0; u < 8; u++) x[u] = term(e + u);
#pragma unroll
                for (int u = 0; u < 8; u++) acc += x[u];
            }
            for (; e < s1; e++) acc += term(e);
            gi[(size_t)t * c + ch] = sign * acc;
        }
    }
}

// g1[p, ch] += sum over s of go[p, s, ch]   (the per-row half of K8: no table needed, fixed order)
__global__ __launch_bounds__(256) void rows_sum_kernel(long long total, int ns, int c, const float* __restrict__ go, float* __restrict__ g1)
{
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long p = e / c; const int ch = (int)(e - p * c);
        float acc = g1[e];
        for (int s = 0; s < ns; s++) acc += go[((size_t)p * ns + s) * c + ch];
        g1[e] = acc;
    }
}

}  // namespace

CBL_EXPORT size_t cbl_neighbor_transpose_workspace_bytes(int m, int n, int nsample)
{
    if (m < 0 || n < 0 || nsample <= 0) return 0;
    NtWs w;
    return nt_carve(w, nullptr, m, n, nsample);
}

CBL_EXPORT int cbl_neighbor_transpose(int m, int n, int nsample, const int* idx, const int* order_src, const int* order_dst, int* inv_start, int* inv_src,
                                      void* workspace, size_t workspace_bytes, void* stream)
{
    if (m < 0 || n < 0 || nsample <= 0 || !inv_start) return CBL_ERR_BAD_ARG;
    if ((long long)m * nsample > 0x7fffffffLL) return CBL_ERR_BAD_ARG;
    hipStream_t st = cbl_stream(stream);
    const int ntt = nt_tiles(n), nst = nt_src_tiles(m);
    if (m == 0 || n == 0) {
        if (hipMemsetAsync(inv_start, 0, sizeof(int) * (size_t)(n + 1), st) != hipSuccess) return cbl_status();
        return CBL_OK;
    }
    if (!idx || !inv_src || !workspace) return CBL_ERR_BAD_ARG;
    if (ntt > NT_MAX_TILES) return CBL_ERR_UNSUPPORTED;              // n > 1 M points: the scatter kernels remain
    NtWs w;
    if (nt_carve(w, (char*)workspace, m, n, nsample) > workspace_bytes) return CBL_ERR_WORKSPACE;
    const CblFastDiv dv = cbl_fastdiv_make((unsigned)nsample);
    const int nzero = ntt + 2;
    hipLaunchKernelGGL(nt_prep_kernel, dim3(cbl_div_up(n > nzero ? n : nzero, NB)), dim3(NB), 0, st, n, nzero, order_dst, w.rank, w.tile_cursor, (int*)nullptr);
    const int* rank = order_dst ? w.rank : nullptr;
    const long long cap = (long long)TS * nsample < ntt ? (long long)TS * nsample : ntt;
    const size_t lds_count = sizeof(int) * (size_t)((ntt + 1) & ~1) + sizeof(int2) * (size_t)cap, lds_bin = sizeof(int) * 2 * (size_t)ntt;
#define CBL_NT(UN_)                                                                                                                                   \
    hipLaunchKernelGGL((nt_count_kernel<UN_>), dim3(nst), dim3(NB), lds_count, st, m, n, nsample, ntt, dv, idx, order_src, rank, w.tile_cursor,       \
                       w.list_cursor, w.tile_list_off, w.tile_list_n, w.lists);                                                                       \
    if (ntt > NT_SCAN_SPLIT) {                                                                                                                        \
        hipLaunchKernelGGL(nt_scan_kernel, dim3(1), dim3(1024), 0, st, ntt, w.tile_cursor, w.tile_base);                                              \
        hipLaunchKernelGGL((nt_bin_kernel<UN_, true>), dim3(nst), dim3(NB), lds_bin, st, m, n, nsample, ntt, dv, idx, order_src, rank, w.tile_cursor, \
                           w.tile_list_off, w.tile_list_n, w.lists, w.tile_base, w.bins);                                                             \
    } else                                                                                                                                            \
        hipLaunchKernelGGL((nt_bin_kernel<UN_, false>), dim3(nst), dim3(NB), lds_bin, st, m, n, nsample, ntt, dv, idx, order_src, rank, w.tile_cursor, \
                           w.tile_list_off, w.tile_list_n, w.lists, w.tile_base, w.bins)
    const int per_thread = (TS * nsample + NB - 1) / NB;             // pairs per thread of a full source tile: one batch where it fits 16
    if (per_thread <= 4) { CBL_NT(4); } else if (per_thread <= 8) { CBL_NT(8); } else { CBL_NT(16); }
#undef CBL_NT
    hipLaunchKernelGGL(nt_finish_kernel, dim3(ntt), dim3(NB), 0, st, n, ntt, w.tile_base, w.bins, w.scratch, inv_start, inv_src);
    return cbl_status();
}

// Two neighbour tables of ONE geometry (same sources, same targets, same orders: the block's K = 8 / 16 table and the CBL head's K = 36 table of a stage) transposed
// by the same four launches: the values are those of two cbl_neighbor_transpose calls — the same kernels' bodies, blockIdx.y picks the table, the rank array is shared.
// Workspace: the second table's scratch behind the first's.
CBL_EXPORT size_t cbl_neighbor_transpose_pair_workspace_bytes(int m, int n, int nsample_a, int nsample_b)
{
    if (m < 0 || n < 0 || nsample_a <= 0 || nsample_b <= 0) return 0;
    NtWs w;
    return nt_carve(w, nullptr, m, n, nsample_a) + nt_carve(w, nullptr, m, n, nsample_b);
}

CBL_EXPORT int cbl_neighbor_transpose_pair(int m, int n, int nsample_a, const int* idx_a, int nsample_b, const int* idx_b, const int* order_src, const int* order_dst,
                                           int* inv_start_a, int* inv_src_a, int* inv_start_b, int* inv_src_b, void* workspace, size_t workspace_bytes, void* stream)
{
    if (m < 0 || n < 0 || nsample_a <= 0 || nsample_b <= 0 || !inv_start_a || !inv_start_b) return CBL_ERR_BAD_ARG;
    if ((long long)m * nsample_a > 0x7fffffffLL || (long long)m * nsample_b > 0x7fffffffLL) return CBL_ERR_BAD_ARG;
    const int ntt = nt_tiles(n), nst = nt_src_tiles(m);
    NtWs wa, wb;
    const size_t bytes_a = nt_carve(wa, (char*)workspace, m, n, nsample_a);
    if (m == 0 || n == 0 || ntt > NT_SCAN_SPLIT) {
        // empty tables, or tables large enough for the split scan: one after the other (second table's scratch behind the first's, as below)
        if (m > 0 && n > 0 && (!workspace || bytes_a + nt_carve(wb, nullptr, m, n, nsample_b) > workspace_bytes)) return workspace ? CBL_ERR_WORKSPACE : CBL_ERR_BAD_ARG;
        const int rc = cbl_neighbor_transpose(m, n, nsample_a, idx_a, order_src, order_dst, inv_start_a, inv_src_a, workspace, bytes_a, stream);
        if (rc) return rc;
        return cbl_neighbor_transpose(m, n, nsample_b, idx_b, order_src, order_dst, inv_start_b, inv_src_b, workspace ? (char*)workspace + bytes_a : nullptr,
                                      workspace_bytes > bytes_a ? workspace_bytes - bytes_a : 0, stream);
    }
    if (!idx_a || !idx_b || !inv_src_a || !inv_src_b || !workspace) return CBL_ERR_BAD_ARG;
    if (bytes_a + nt_carve(wb, (char*)workspace + bytes_a, m, n, nsample_b) > workspace_bytes) return CBL_ERR_WORKSPACE;
    hipStream_t st = cbl_stream(stream);
    auto job = [&](const NtWs& w, int ns, const int* idx, int* inv_start, int* inv_src) {
        NtJob j;
        j.ns = ns; j.dv = cbl_fastdiv_make((unsigned)ns); j.idx = idx;
        j.tile_cursor = w.tile_cursor; j.list_cursor = w.list_cursor; j.tile_list_off = w.tile_list_off; j.tile_list_n = w.tile_list_n; j.tile_base = w.tile_base;
        j.lists = w.lists; j.bins = w.bins; j.scratch = w.scratch; j.inv_start = inv_start; j.inv_src = inv_src;
        return j;
    };
    // the wider table first: the launch's template arguments are dispatched on (wide, narrow) batch sizes
    const bool swap = nsample_b > nsample_a;
    const NtJob ja = swap ? job(wb, nsample_b, idx_b, inv_start_b, inv_src_b) : job(wa, nsample_a, idx_a, inv_start_a, inv_src_a);
    const NtJob jb = swap ? job(wa, nsample_a, idx_a, inv_start_a, inv_src_a) : job(wb, nsample_b, idx_b, inv_start_b, inv_src_b);
    const int nzero = ntt + 2;
    hipLaunchKernelGGL(nt_prep_kernel, dim3(cbl_div_up(n > nzero ? n : nzero, NB)), dim3(NB), 0, st, n, nzero, order_dst, wa.rank, wa.tile_cursor, wb.tile_cursor);
    const int* rank = order_dst ? wa.rank : nullptr;
    const int ns_max = ja.ns;
    const long long cap = (long long)TS * ns_max < ntt ? (long long)TS * ns_max : ntt;
    const size_t lds_count = sizeof(int) * (size_t)((ntt + 1) & ~1) + sizeof(int2) * (size_t)cap, lds_bin = sizeof(int) * 2 * (size_t)ntt;
    auto un_of = [](int ns) { const int per_thread = (TS * ns + NB - 1) / NB; return per_thread <= 4 ? 4 : per_thread <= 8 ? 8 : 16; };
    const int una = un_of(ja.ns), unb = un_of(jb.ns);
    const dim3 gsrc(nst, 2), gdst(ntt, 2);
#define CBL_NTP(UA_, UB_)                                                                                                                        \
    { hipLaunchKernelGGL((nt_count_pair_kernel<UA_, UB_>), gsrc, dim3(NB), lds_count, st, m, n, ntt, order_src, rank, ja, jb);                  \
      hipLaunchKernelGGL((nt_bin_pair_kernel<UA_, UB_>), gsrc, dim3(NB), lds_bin, st, m, n, ntt, order_src, rank, ja, jb); }
    if (una == 16 && unb == 16) CBL_NTP(16, 16) else if (una == 16 && unb == 8) CBL_NTP(16, 8) else if (una == 16) CBL_NTP(16, 4)
    else if (una == 8 && unb == 8) CBL_NTP(8, 8) else if (una == 8) CBL_NTP(8, 4) else CBL_NTP(4, 4)
#undef CBL_NTP
    hipLaunchKernelGGL(nt_finish_pair_kernel, gdst, dim3(NB), 0, st, n, ntt, ja, jb);
    return cbl_status();
}

CBL_EXPORT int cbl_grouping_backward_csr(int n, int c, const float* grad_output, const int* order_dst, const int* inv_start, const int* inv_src,
                                         float* grad_input, void* stream)
{
    if (n < 0 || c <= 0) return CBL_ERR_BAD_ARG;
    if (n == 0) return CBL_OK;
    if (!grad_output || !inv_start || !inv_src || !grad_input) return CBL_ERR_BAD_ARG;
    hipStream_t st = cbl_stream(stream);
    const bool v4 = (c % 4 == 0) && cbl_host_aligned16(grad_output) && cbl_host_aligned16(grad_input);
    const int lg = c / 4;
#define CBL_GBC(LG_) { const unsigned nwg = cbl_div_up(n, NB / LG_); unsigned g = cbl_round_up8(nwg); if (g > 256u * 16u) g = 256u * 16u;  \
        hipLaunchKernelGGL((grouping_bwd_csr_kernel<LG_>), dim3(g), dim3(NB), 0, st, (unsigned)n, reinterpret_cast<const float4*>(grad_output), order_dst,  \
                           inv_start, inv_src, reinterpret_cast<float4*>(grad_input)); return cbl_status(); }
    if (v4) switch (lg) {
        case 1: CBL_GBC(1) case 2: CBL_GBC(2) case 4: CBL_GBC(4) case 8: CBL_GBC(8) case 16: CBL_GBC(16) case 32: CBL_GBC(32) case 64: CBL_GBC(64)
        default: break;
    }
#undef CBL_GBC
    hipLaunchKernelGGL(grouping_bwd_csr_scalar_kernel, dim3(cbl_grid_for((long long)n * c, NB)), dim3(NB), 0, st, (unsigned)n, c, grad_output, order_dst, inv_start,
                       inv_src, grad_input);
    return cbl_status();
}

CBL_EXPORT int cbl_grouping_backward_csr_rows(int n, int c, int row_stride, int col_offset, const float* grad_output, const int* order_dst,
                                              const int* inv_start, const int* inv_src, float* grad_input, void* stream)
{
    if (n < 0 || c <= 0 || col_offset < 0 || row_stride < col_offset + c) return CBL_ERR_BAD_ARG;
    if (n == 0) return CBL_OK;
    if (!grad_output || !inv_start || !inv_src || !grad_input) return CBL_ERR_BAD_ARG;
    if (row_stride == c && col_offset == 0) return cbl_grouping_backward_csr(n, c, grad_output, order_dst, inv_start, inv_src, grad_input, stream);
    unsigned g = cbl_round_up8(cbl_div_up(n, 4)); if (g > 256u * 32u) g = 256u * 32u;
    hipLaunchKernelGGL(grouping_bwd_csr_rows_kernel, dim3(g), dim3(NB), 0, cbl_stream(stream), (unsigned)n, c, row_stride, col_offset, grad_output, order_dst,
                       inv_start, inv_src, grad_input);
    return cbl_status();
}

// K6 / the grad_input part of K10 as a gather: grad_input[t, ch] = sum over the pairs p = (source, column) that list t, ascending, of
// rows[source, ch] * weight[p, ch % w_c]   (K6: nsample = 3, w_c = 1, interpolation_cuda_kernel.cu:20-33; K10: aggregation_cuda_kernel.cu:22-39).
// grad_input is written, not accumulated.
CBL_EXPORT int cbl_weighted_scatter_csr(int n, int nsample, int c, int w_c, const float* rows, const float* weight, const int* order_dst,
                                        const int* inv_start, const int* inv_src, float* grad_input, void* stream)
{
    if (n < 0 || nsample <= 0 || c <= 0 || w_c <= 0) return CBL_ERR_BAD_ARG;
    if (n == 0) return CBL_OK;
    if (!rows || !weight || !inv_start || !inv_src || !grad_input) return CBL_ERR_BAD_ARG;
    unsigned g = cbl_round_up8(cbl_div_up(n, 4)); if (g > 256u * 32u) g = 256u * 32u;
    hipLaunchKernelGGL((scatter_as_gather_kernel<false, true>), dim3(g), dim3(NB), 0, cbl_stream(stream), (unsigned)n, c, cbl_fastdiv_make((unsigned)nsample), w_c, 1.0f,
                       rows, weight, order_dst, inv_start, inv_src, grad_input);
    return cbl_status();
}

// K8 (subtraction_cuda_kernel.cu:18-30) without atomics: grad_input1[p] += sum_s grad_output[p, s] (accumulated, as the reference does into its
// zero-filled tensor), grad_input2[t] = -(sum over the pairs that list t, ascending) (written)
CBL_EXPORT int cbl_subtraction_backward_csr(int m, int n2, int nsample, int c, const float* grad_output, const int* order_dst, const int* inv_start,
                                            const int* inv_src, float* grad_input1, float* grad_input2, void* stream)
{
    if (m < 0 || n2 < 0 || nsample <= 0 || c <= 0) return CBL_ERR_BAD_ARG;
    if (!grad_output || !inv_start || !inv_src || !grad_input1 || !grad_input2) return CBL_ERR_BAD_ARG;
    hipStream_t st = cbl_stream(stream);
    if (m > 0)
        hipLaunchKernelGGL(rows_sum_kernel, dim3(cbl_grid_for((long long)m * c, 256)), dim3(256), 0, st, (long long)m * c, nsample, c, grad_output, grad_input1);
    if (n2 > 0) {
        unsigned g = cbl_round_up8(cbl_div_up(n2, 4)); if (g > 256u * 32u) g = 256u * 32u;
        hipLaunchKernelGGL((scatter_as_gather_kernel<true, false>), dim3(g), dim3(NB), 0, st, (unsigned)n2, c, cbl_fastdiv_make(1u), 1, -1.0f,
                           grad_output, nullptr, order_dst, inv_start, inv_src, grad_input2);
    }
    return cbl_status();
}
