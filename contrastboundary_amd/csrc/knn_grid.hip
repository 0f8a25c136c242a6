// K1 (fast): uniform-grid KNN with tie certification and exact replay.
// Same results as knnquery_cuda_kernel (/root/reference/pytorch/lib/pointops/src/knnquery/knnquery_cuda_kernel.cu:65-111),
// bit for bit, in O(m * candidates) instead of O(m * n):
//   1. per cloud: bounding box -> cell size -> dense grid (cbl_grid_choose, grid_core.h)
//   2. counting sort of the supports by cell (histogram with L2 atomics, one-workgroup scan, scatter)
//      into float4 {x,y,z,bits(idx)} so a neighbourhood row of cells is ONE contiguous, coalescable range
//   3. one 16-lane GROUP per query (64-lane for 16 < K <= 64), queries taken in cell-sorted order when the
//      query set is the support set (the 4 groups of a wave then walk the same cells -> L1/L2 hits).  The
//      group's lanes fetch 16 consecutive candidates of a cell row with one coalesced 256 B load, and the
//      ascending top-K list is DISTRIBUTED over the lanes (element j in lane j): inserting a candidate is
//      one compare + one DPP row-shift, independent of K.  (A lane-per-query version of the same search,
//      cbl_knn_grid_query in grid_core.h, is what tests/host_emul runs on the CPU.)
//   4. a query is final only if its result provably does not depend on the reference's heap layout:
//      K distinct distances and no outside candidate tied with the K-th (CblTopK::certify) — everything else
//      (ties, clouds with <= K supports) is appended to a worklist and recomputed by the exact kernel
//      (knn_exact.hip) in the reference's visiting order.  No host synchronisation anywhere.
#include "cbl_common.h"
#include <cstdlib>
#include "grid_core.h"
#include <knn_wave.h>

int cbl_knn_exact_worklist(int b, int m, int K, const float* xyz, const float* new_xyz, const int* offset,
                           const int* new_offset, int* idx, float* dist2,
                           const int* worklist, const int* worklist_count, int max_work, hipStream_t st,
                           const void* grids, const int* cell_start, const void* sorted);

namespace {

constexpr int GRID_MAX_K = 64;          // larger nsample -> exact kernel (register list would spill)
constexpr int CELLS_PER_POINT = 2;      // cell capacity per cloud = 2*n_c + 64
constexpr int CELLS_PER_CLOUD = 64;

struct Workspace {
    CblGrid* grids;      // [b]
    int* counters;       // [0] = worklist length
    unsigned* bbox;      // [6*b] order-preserving keys of the per-cloud min / max
    int* cell_count;     // [ncap + 1]  histogram, then running fill cursor
    int* cell_start;     // [ncap + 1]  exclusive scan
    int* cell_local;     // [ncap + 1]  tile-local exclusive scan (intermediate)
    int* tile_sum;       // [ceil((ncap + 1) / 4096)]
    int* pt_cell;        // [n]
    float4* sorted;      // [n]
    int* worklist;       // [m]
    int* worklist2;      // [m]  tied rows of the narrower result that rides along with a wave-kernel search (cbl_knnquery_nested)
    size_t bytes;
    int ncap;
};

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

Workspace carve(void* base, int b, int n, int m)
{
    Workspace w;
    w.ncap = CELLS_PER_POINT * n + CELLS_PER_CLOUD * b;
    char* p = reinterpret_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) { char* r = p ? p + off : nullptr; off += align_up(bytes); return r; };
    w.grids = reinterpret_cast<CblGrid*>(take(sizeof(CblGrid) * (size_t)b));
    w.counters = reinterpret_cast<int*>(take(sizeof(int) * 64));
    w.bbox = reinterpret_cast<unsigned*>(take(sizeof(unsigned) * 6 * (size_t)b));
    w.cell_count = reinterpret_cast<int*>(take(sizeof(int) * ((size_t)w.ncap + 1)));
    w.cell_start = reinterpret_cast<int*>(take(sizeof(int) * ((size_t)w.ncap + 1)));
    w.cell_local = reinterpret_cast<int*>(take(sizeof(int) * ((size_t)w.ncap + 1)));
    w.tile_sum = reinterpret_cast<int*>(take(sizeof(int) * ((size_t)w.ncap / 4096 + 2)));
    w.pt_cell = reinterpret_cast<int*>(take(sizeof(int) * (size_t)n));
    w.sorted = reinterpret_cast<float4*>(take(sizeof(float4) * (size_t)n));
    w.worklist = reinterpret_cast<int*>(take(sizeof(int) * (size_t)m));
    w.worklist2 = reinterpret_cast<int*>(take(sizeof(int) * (size_t)m));
    w.bytes = off;
    return w;
}

// ---- 1. per-cloud bounding box (many workgroups, L2 atomics on order-preserving keys) + grid parameters ----
__device__ __forceinline__ unsigned f2key(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float key2f(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// bbox[c*6 + a] = min key, bbox[c*6 + 3 + a] = max key; pre-set to 0xffffffff / 0 by the memset below
__global__ __launch_bounds__(256) void grid_bbox_kernel(int b, int n, const float* __restrict__ xyz, const int* __restrict__ offset,
                                                        unsigned* __restrict__ bbox)
{
    // every workgroup takes a contiguous slice of rows.  A slice normally lies inside one cloud; where it
    // straddles a boundary the wave works cloud by cloud: the lowest lane with rows left names the cloud,
    // lanes whose next row is in another cloud sit the round out (so the cross-lane reduction never mixes clouds)
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * per, r1 = min(n, r0 + per);
    const int lane = threadIdx.x & 63;
    int i = r0 + threadIdx.x;
    if (r0 >= r1) return;
    {   // common case: the whole slice lies in one cloud -> reduce across the workgroup in LDS, 6 atomics per workgroup
        const int cfirst = cbl_cloud_of(r0, offset, b);
        if (cfirst == cbl_cloud_of(r1 - 1, offset, b)) {
            __shared__ float red[6][4];
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (; i < r1; i += 256) {
#pragma unroll
                for (int a = 0; a < 3; a++) { const float v = xyz[3 * i + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
            }
#pragma unroll
            for (int a = 0; a < 3; a++) {
                for (int s = 32; s >= 1; s >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], s)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], s)); }
                if (lane == 0) { red[a][threadIdx.x >> 6] = lo[a]; red[3 + a][threadIdx.x >> 6] = hi[a]; }
            }
            __syncthreads();
            if (threadIdx.x < 6) {
                const int a = threadIdx.x;
                const float v0 = red[a][0], v1 = red[a][1], v2 = red[a][2], v3 = red[a][3];
                if (a < 3) atomicMin(bbox + cfirst * 6 + a, f2key(fminf(fminf(v0, v1), fminf(v2, v3))));
                else       atomicMax(bbox + cfirst * 6 + a, f2key(fmaxf(fmaxf(v0, v1), fmaxf(v2, v3))));
            }
            return;
        }
    }
    while (__any(i < r1)) {
        const bool act = i < r1;
        const int myc = act ? cbl_cloud_of(i, offset, b) : -1;
        const int lead = __builtin_ctzll(__ballot(act));
        const int c0 = __shfl(myc, lead);
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        if (act && myc == c0) {
            const int cend = min(r1, offset[c0]);
            for (; i < cend; i += 256) {
#pragma unroll
                for (int a = 0; a < 3; a++) { const float v = xyz[3 * i + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
            }
        }
#pragma unroll
        for (int a = 0; a < 3; a++) {
            for (int s = 32; s >= 1; s >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], s)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], s)); }
            // 6 atomics per wave round: ~1000 atomics on 6 addresses for a 40960-point cloud; they serialise in L2 but cost
            // less than a cross-wave LDS stage would for clouds that straddle workgroups
            if (lane == 0 && lo[a] <= hi[a]) { atomicMin(bbox + c0 * 6 + a, f2key(lo[a])); atomicMax(bbox + c0 * 6 + 3 + a, f2key(hi[a])); }
        }
    }
}

__device__ __forceinline__ CblGrid grid_of_cloud(int c, float pts_per_cell, const int* __restrict__ offset, const unsigned* __restrict__ bbox)
{
    const int start = c ? offset[c - 1] : 0, end = offset[c];
    float l[3] = {0.f, 0.f, 0.f}, h[3] = {0.f, 0.f, 0.f};
    if (end > start)
        for (int a = 0; a < 3; a++) { l[a] = key2f(bbox[c * 6 + a]); h[a] = key2f(bbox[c * 6 + 3 + a]); }
    CblGrid g;
    const int cap = CELLS_PER_POINT * (end - start) + CELLS_PER_CLOUD;
    if (pts_per_cell < 0.f) cbl_grid_choose_radius(g, l, h, -pts_per_cell, cap);      // radius search: cell edge ~ radius
    else cbl_grid_choose(g, l, h, end - start, pts_per_cell, cap);
    g.cell_base = CELLS_PER_POINT * start + CELLS_PER_CLOUD * c;
    g.start = start; g.end = end; g.pad0 = g.pad1 = 0;
    return g;
}

// ---- 2a. grid parameters (derived from the bounding boxes by every workgroup, published by workgroup 0) + histogram ----
__global__ __launch_bounds__(256) void grid_count_kernel(int b, int n, float pts_per_cell, const float* __restrict__ xyz, const int* __restrict__ offset,
                                                         const unsigned* __restrict__ bbox, CblGrid* __restrict__ grids,
                                                         int* __restrict__ pt_cell, int* __restrict__ cell_count)
{
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < b; c += 256) grids[c] = grid_of_cloud(c, pts_per_cell, offset, bbox);
    int cached_c = -1; CblGrid g;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int c = cbl_cloud_of(i, offset, b);
        if (c != cached_c) { g = grid_of_cloud(c, pts_per_cell, offset, bbox); cached_c = c; }
        const int cell = cbl_cell_of(g, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
        pt_cell[i] = cell;
        atomicAdd(cell_count + cell, 1);
    }
}

// ---- 2b. exclusive scan of the histogram: tile-local scans (one workgroup per 4096 entries, 16 per lane as
// four independent 16 B loads) + a second launch that adds each tile's prefix.  Leaves the running fill cursor
// in cell_count.  (A single-workgroup scan is latency-bound: ~1 us per dependent L2 round trip.)
constexpr int SCAN_TILE = 4096;

__global__ __launch_bounds__(256) void grid_scan_tiles_kernel(int total, const int* __restrict__ cell_count, int* __restrict__ cell_start_local,
                                                              int* __restrict__ tile_sum)
{
    __shared__ int wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = blockIdx.x * SCAN_TILE + tid * 16;
    int v[16];
#pragma unroll
    for (int j = 0; j < 16; j += 4) {
        if (base + j + 3 < total) {
            const int4 x = *reinterpret_cast<const int4*>(cell_count + base + j);
            v[j] = x.x; v[j + 1] = x.y; v[j + 2] = x.z; v[j + 3] = x.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) v[j + k] = (base + j + k < total) ? cell_count[base + j + k] : 0;
        }
    }
    int sum = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) { const int x = v[j]; v[j] = sum; sum += x; }     // thread-local exclusive
    int incl = sum;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) { const int o = __shfl_up(incl, s); if (lane >= s) incl += o; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int off = incl - sum;
    for (int w = 0; w < wave; w++) off += wsum[w];
    if (tid == 255) tile_sum[blockIdx.x] = off + sum;
#pragma unroll
    for (int j = 0; j < 16; j++) if (base + j < total) cell_start_local[base + j] = off + v[j];
}

// ---- 2c. finish the scan (add each tile's prefix -> final cell_start) and scatter the supports into cell order ---------
// Slot of a point = cell_start[cell] + (atomicSub(count[cell]) - 1): the histogram is consumed instead of a separate cursor array
// (the order inside a cell is irrelevant) and ends at zero, ready for the next build.
__global__ __launch_bounds__(256) void grid_scatter_kernel(int n, int total, int ntiles, const float* __restrict__ xyz, const int* __restrict__ pt_cell,
                                                           const int* __restrict__ tile_sum, int* __restrict__ cell_start_local, int* __restrict__ cell_start,
                                                           int* __restrict__ cell_count, float4* __restrict__ sorted, int* __restrict__ order_out)
{
    extern __shared__ int tile_lds[];                      // [0, ntiles): tile totals, [ntiles, 2*ntiles): their exclusive prefix
    int* tile_pre = tile_lds + ntiles;
    {
        // every workgroup rebuilds the (short) prefix of the tile totals itself: cheaper than another launch
        for (int t = threadIdx.x; t < ntiles; t += 256) tile_lds[t] = tile_sum[t];       // independent loads, all in flight
        __syncthreads();
        for (int t = threadIdx.x; t < ntiles; t += 256) { int run = 0; for (int u = 0; u < t; u++) run += tile_lds[u]; tile_pre[t] = run; }
        __syncthreads();
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) cell_start[i] = cell_start_local[i] + tile_pre[i / SCAN_TILE];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int cell = pt_cell[i];
        const int pos = cell_start_local[cell] + tile_pre[cell / SCAN_TILE] + atomicSub(cell_count + cell, 1) - 1;
        sorted[pos] = make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], __int_as_float(i));
        if (order_out) order_out[pos] = i;                  // the cell order of the supports, for the consumers' processing order (cbl_knnquery_ordered)
    }
}

// ---- 2d. the exported cell order made run-to-run deterministic: inside a cell the scatter above places points in the order their
// atomics happen to retire.  No search result depends on that, and neither does any VALUE downstream — except through rounding where a
// consumer SUMS in processing order (the fused attention layer's BatchNorm statistics, pt_layer.hip): the same step then differed from run
// to run in the last bit, which a few SGD steps amplify (measured: 6e-7 on the loss at step 0, 3e-2 at step 3).  One thread per point: its
// rank among the ids of its cell (read from the scattered copies: ~15 independent loads) is its slot.  Cells beyond 1024 points (degenerate
// clouds) keep the scatter's order.  (A thread per CELL sorting its segment in place was 350 us: a chain of dependent global accesses.)
__device__ __forceinline__ void grid_order_canon_body(int n, const int* __restrict__ pt_cell, const int* __restrict__ cell_start,
                                                      const float4* __restrict__ sorted, int* __restrict__ order, int block, int nblocks)
{
    for (int i = block * 256 + threadIdx.x; i < n; i += nblocks * 256) {
        const int cell = pt_cell[i];
        const int s = cell_start[cell], e = cell_start[cell + 1];
        if (e - s < 2 || e - s > 1024) continue;
        int rank = 0;
        for (int j = s; j < e; j += 8) {                          // eight independent loads per trip (entries past the end: the last one again, not counted)
            int v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = __float_as_int(sorted[min(j + u, e - 1)].w);
#pragma unroll
            for (int u = 0; u < 8; u++) rank += (j + u < e && v[u] < i) ? 1 : 0;
        }
        order[s + rank] = i;
    }
}
__global__ __launch_bounds__(256) void grid_order_canon_kernel(int n, const int* __restrict__ pt_cell, const int* __restrict__ cell_start,
                                                               const float4* __restrict__ sorted, int* __restrict__ order)
{
    grid_order_canon_body(n, pt_cell, cell_start, sorted, order, blockIdx.x, gridDim.x);
}
// the same pass riding in the search launch (knn_grid_wave_kernel's first `canon_blocks` workgroups): nothing of the search reads the exported order, so the
// ~8 us of dependent loads run beside the queries instead of in a launch of their own in front of them
struct CblCanon { int blocks, n; const int* pt_cell; int* order; };

// ---- 3. queries: one G-lane group per query ------------------------------------------------------------------
template <int G> __device__ __forceinline__ float dpp_shr1_f(float v);
template <int G> __device__ __forceinline__ int dpp_shr1_i(int v);
// row_shr:1 (0x111) shifts inside each 16-lane row, wave_shr:1 (0x138) across the whole wave; lane 0 of the
// row / wave keeps `old` (= its own value, masked out by the caller)
template <> __device__ __forceinline__ int dpp_shr1_i<16>(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false); }
template <> __device__ __forceinline__ int dpp_shr1_i<64>(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }
template <> __device__ __forceinline__ int dpp_shr1_i<32>(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }   // lane 32 receives lane 31: masked by gl > 0
// the same shifts with bound_ctrl: the first lane of the row / wave receives 0 (the bit pattern of +0.0: below every distance)
template <int G> __device__ __forceinline__ int dpp_shr1_zero(int v)
{
    if constexpr (G == 16) return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    else                   return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true);
}
template <> __device__ __forceinline__ float dpp_shr1_f<16>(float v) { return __int_as_float(dpp_shr1_i<16>(__float_as_int(v))); }
template <> __device__ __forceinline__ float dpp_shr1_f<64>(float v) { return __int_as_float(dpp_shr1_i<64>(__float_as_int(v))); }
template <> __device__ __forceinline__ float dpp_shr1_f<32>(float v) { return __int_as_float(dpp_shr1_i<32>(__float_as_int(v))); }

constexpr int WV_NB = 16;                                           // candidate registers per lane: T <= 1024

template <int CTRL> __device__ __forceinline__ int dppx_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }   // every lane has a source: `old` is never used
struct PermQuadXor1  { __device__ __forceinline__ int operator()(int v, int) const { return dppx_i<0xB1>(v); } };    // lane ^ 1
struct PermQuadXor2  { __device__ __forceinline__ int operator()(int v, int) const { return dppx_i<0x4E>(v); } };    // lane ^ 2
struct PermQuadMir   { __device__ __forceinline__ int operator()(int v, int) const { return dppx_i<0x1B>(v); } };    // lane ^ 3
struct PermHalfMir   { __device__ __forceinline__ int operator()(int v, int) const { return dppx_i<0x141>(v); } };   // lane ^ 7
struct PermRowMir    { __device__ __forceinline__ int operator()(int v, int) const { return dppx_i<0x140>(v); } };   // lane ^ 15
struct PermRowRor8   { __device__ __forceinline__ int operator()(int v, int) const { return dppx_i<0x128>(v); } };   // lane ^ 8
struct PermSwzXor4   { __device__ __forceinline__ int operator()(int v, int) const { return __builtin_amdgcn_ds_swizzle(v, 0x101f); } };
struct PermSwzXor16  { __device__ __forceinline__ int operator()(int v, int) const { return __builtin_amdgcn_ds_swizzle(v, 0x401f); } };
struct PermSwzMir32  { __device__ __forceinline__ int operator()(int v, int) const { return __builtin_amdgcn_ds_swizzle(v, 0x7c1f); } };   // lane ^ 31
struct PermMir64     { __device__ __forceinline__ int operator()(int v, int lane) const { return __builtin_amdgcn_ds_bpermute((63 - lane) << 2, v); } };

// compare-exchange with the partner lane: the lower lane of a pair keeps the smaller (distance, index)
template <bool LEX, class Perm> __device__ __forceinline__ void sort_cx(float& d, int& i, bool lower, int lane, Perm perm)
{
    const float pd = __int_as_float(perm(__float_as_int(d), lane));
    const int pi = perm(i, lane);
    if (LEX) {
        const bool take = lower ? (pd < d || (pd == d && (unsigned)pi < (unsigned)i)) : (pd > d || (pd == d && (unsigned)pi > (unsigned)i));
        d = take ? pd : d; i = take ? pi : i;
    } else {
        // min / max + "did my distance change" on the BIT PATTERNS (distances are >= +0, so they order like integers; integer
        // min / max need no NaN canonicalisation): 5-7 VALU per stage instead of the 13 of a ?: over two float compares.
        // On a tie both lanes keep their own (distance, index).
        const int di = __float_as_int(d), pdi = __float_as_int(pd);
        const int nd = lower ? min(di, pdi) : max(di, pdi);
        i = (nd != di) ? pi : i; d = __int_as_float(nd);
    }
}
// ascending bitonic sort of one (d, i) per lane over the 64 lanes ("flip" form: every merge starts with a mirror exchange)
template <bool LEX> __device__ __forceinline__ void wave_sort64(float& d, int& i, int lane)
{
    const bool l1 = !(lane & 1), l2 = !(lane & 2), l4 = !(lane & 4), l8 = !(lane & 8), l16 = !(lane & 16), l32 = !(lane & 32);
    sort_cx<LEX>(d, i, l1, lane, PermQuadXor1());
    sort_cx<LEX>(d, i, l2, lane, PermQuadMir());   sort_cx<LEX>(d, i, l1, lane, PermQuadXor1());
    sort_cx<LEX>(d, i, l4, lane, PermHalfMir());   sort_cx<LEX>(d, i, l2, lane, PermQuadXor2()); sort_cx<LEX>(d, i, l1, lane, PermQuadXor1());
    sort_cx<LEX>(d, i, l8, lane, PermRowMir());    sort_cx<LEX>(d, i, l4, lane, PermSwzXor4());  sort_cx<LEX>(d, i, l2, lane, PermQuadXor2());
    sort_cx<LEX>(d, i, l1, lane, PermQuadXor1());
    sort_cx<LEX>(d, i, l16, lane, PermSwzMir32()); sort_cx<LEX>(d, i, l8, lane, PermRowRor8());  sort_cx<LEX>(d, i, l4, lane, PermSwzXor4());
    sort_cx<LEX>(d, i, l2, lane, PermQuadXor2());  sort_cx<LEX>(d, i, l1, lane, PermQuadXor1());
    sort_cx<LEX>(d, i, l32, lane, PermMir64());    sort_cx<LEX>(d, i, l16, lane, PermSwzXor16()); sort_cx<LEX>(d, i, l8, lane, PermRowRor8());
    sort_cx<LEX>(d, i, l4, lane, PermSwzXor4());   sort_cx<LEX>(d, i, l2, lane, PermQuadXor2()); sort_cx<LEX>(d, i, l1, lane, PermQuadXor1());
}
__device__ __forceinline__ float rl_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
// minimum over the G lanes of a group of a value >= +0 (or +inf), on its bit pattern: four row-local DPP steps (the partner's value
// folds into v_min_i32 as a DPP operand), then across the rows; every lane of the group gets the result
template <int G> __device__ __forceinline__ float group_min_nonneg(float f)
{
    int v = __float_as_int(f);
    v = min(v, dppx_i<0xB1>(v)); v = min(v, dppx_i<0x4E>(v)); v = min(v, dppx_i<0x141>(v)); v = min(v, dppx_i<0x140>(v));
    if constexpr (G == 32) v = min(v, __builtin_amdgcn_ds_swizzle(v, 0x401f));
    if constexpr (G == 64) v = min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
                                   min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
    return __int_as_float(v);
}

// LEX: order the list by (d2, index) instead of d2 alone — needed only by the "any tie" policy (set_exact == 2), whose result must not
// depend on the order in which the cells were filled; the other policies replay every tie that matters and skip the extra compares.
template <int G, bool SELF, bool LEX>
__global__ __launch_bounds__(256) void knn_grid_group_kernel(int b, int m, int K, const float* __restrict__ new_xyz,
                                                             const int* __restrict__ offset, const int* __restrict__ new_offset,
                                                             const CblGrid* __restrict__ grids, const int* __restrict__ cell_start,
                                                             const float4* __restrict__ sorted, int* __restrict__ idx, float* __restrict__ dist2,
                                                             int* __restrict__ worklist, int* __restrict__ counters, int set_exact)
{
    constexpr int QPW = 64 / G;                                     // queries per wave
    using mask_t = unsigned long long;
    const int lane = threadIdx.x & 63;
    const int gl = lane & (G - 1);                                  // lane within the group
    const int grp = lane / G;                                       // group within the wave
    const int wave_global = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int t = wave_global * QPW + grp;                          // group-uniform query slot
    const bool live = t < m;
    const int tt = live ? t : m - 1;                                // dead groups shadow the last query, write nothing

    int q; float qx, qy, qz;
    if (SELF) { const float4 s = sorted[tt]; q = __float_as_int(s.w); qx = s.x; qy = s.y; qz = s.z; }   // cell order
    else      { q = tt; qx = new_xyz[3 * q]; qy = new_xyz[3 * q + 1]; qz = new_xyz[3 * q + 2]; }
    const int c = cbl_cloud_of(q, SELF ? offset : new_offset, b);
    const CblGrid g = grids[c];
    const float uqx = cbl_u(qx, g.ox, g.inv_cs), uqy = cbl_u(qy, g.oy, g.inv_cs), uqz = cbl_u(qz, g.oz, g.inv_cs);
    const int cx = cbl_cell_coord(uqx, g.nx), cy = cbl_cell_coord(uqy, g.ny), cz = cbl_cell_coord(uqz, g.nz);

    float ed = INFINITY; int ei = -1;                               // element `gl` of the group's ascending top list
    float rejmin = INFINITY;                                        // min d2 over candidates not in the list (per lane, reduced at the end)
    int evict = 0x7f800000;                                         // bit pattern; lane K-1's copy is the one that counts (merged below)
    const int last_src = ((lane & ~(G - 1)) + K - 1) << 2;          // ds_bpermute address of the group's K-th list element
    bool done = !live;

    // the support ranges of the 9 rows of the initial block, fetched by lanes 0..8 of the group in ONE round of loads (instead of nine
    // dependent ones, row after row) and handed out by ds_bpermute below.  Row order = the packed "nearest first" table of the loop.
    int rs9 = 0, re9 = 0;
    if (!done && gl < 9) {
        const int o = (int)((0xa82091645ull >> (4 * gl)) & 0xFull);
        const int y = cy + (o & 3) - 1, z = cz + (o >> 2) - 1;
        if (y >= 0 && y < g.ny && z >= 0 && z < g.nz) {
            const int row = g.cell_base + g.nx * (y + g.ny * z);
            rs9 = cell_start[row + max(cx - 1, 0)]; re9 = cell_start[row + min(cx + 1, g.nx - 1) + 1];
        }
    }
    const int grp_src = (lane & ~(G - 1)) << 2;

    for (int r = 1;; r++) {
        // r == 1: the 3x3x3 block around the query's cell (shells 0 and 1) as 9 full rows;
        // r >= 2: shell r — face rows take the whole x-range, interior rows only the two end cells
        // rows of the (2r+1)^2 (dy,dz) square, nearest first for the initial block: the row through the query's own cell, the 4
        // face rows, the 4 corner rows — the list fills with near candidates early, so far rows rarely pass `d2 < worst`
        const int side = 2 * r + 1;
        for (int ri = 0; ri < side * side; ri++) {
            int dy, dz;
            if (r == 1) { const int o = (int)((0xa82091645ull >> (4 * ri)) & 0xFull); dy = (o & 3) - 1; dz = (o >> 2) - 1; }   // packed order table below
            else        { dz = ri / side - r; dy = ri % side - r; }
            {
                const bool full = (r == 1) || dz == -r || dz == r || dy == -r || dy == r;
                const int nseg = full ? 1 : 2;
                for (int sg = 0; sg < nseg; sg++) {
                    int x0, x1;
                    if (full) { x0 = max(cx - r, 0); x1 = min(cx + r, g.nx - 1); }
                    else      { x0 = x1 = sg ? cx + r : cx - r; }
                    const int y = cy + dy, z = cz + dz;
                    int s = 0, e = 0;
                    if (r == 1) {
                        s = __builtin_amdgcn_ds_bpermute(grp_src + (ri << 2), rs9); e = __builtin_amdgcn_ds_bpermute(grp_src + (ri << 2), re9);
                    } else if (!done && y >= 0 && y < g.ny && z >= 0 && z < g.nz && x0 >= 0 && x1 <= g.nx - 1) {
                        const int row = g.cell_base + g.nx * (y + g.ny * z);
                        s = cell_start[row + x0]; e = cell_start[row + x1 + 1];
                    }
                    // ---- the group consumes its range G candidates at a time (wave-level loop: any group busy)
                    for (int p = s; __any(p < e); p += G) {
                        const int pi = p + gl;
                        float d2 = INFINITY; int ci = -1;
                        if (pi < e) {
                            const float4 v = sorted[pi];
                            d2 = cbl_dist2(qx, qy, qz, v.x, v.y, v.z);       // (new - x)^2 ..., knnquery_cuda_kernel.cu:99
                            ci = __float_as_int(v.w);
                        }
                        const float worst = __int_as_float(__builtin_amdgcn_ds_bpermute(last_src, __float_as_int(ed)));
                        const unsigned worst_i = LEX ? (unsigned)__builtin_amdgcn_ds_bpermute(last_src, ei) : 0u;
                        const bool pass = d2 < worst || (LEX && d2 == worst && (unsigned)ci < worst_i);
                        rejmin = __int_as_float(min(__float_as_int(rejmin), pass ? 0x7f800000 : __float_as_int(d2)));   // bit patterns of values >= +0
                        if constexpr (!LEX && G <= 32) {
                            // insertion on the BIT PATTERNS of the distances (all >= +0: they order like integers):
                            //   new e[j] = min(e[j], max(e[j-1], c))  with e[-1] = 0   — 16 VALU per step instead of 36
                            unsigned gm = (unsigned)(__ballot(pass) >> (grp * G)) & (G == 32 ? ~0u : ((1u << G) - 1u));
                            const int d2i = __float_as_int(d2);
                            auto step = [&]() {                               // each group inserts its next passing candidate
                                const bool has = gm != 0;
                                const int l = kw_ffbl(gm);                    // -1 when empty: any lane, masked by `has`
                                gm &= gm - 1;
                                const int src = (l << 2) + grp * (G * 4);
                                int dci = __builtin_amdgcn_ds_bpermute(src, d2i); const int ic = __builtin_amdgcn_ds_bpermute(src, ci);
                                dci = has ? dci : 0x7f800000;
                                const int edi = __float_as_int(ed);
                                int pdi = dpp_shr1_zero<G>(edi), pidx = dpp_shr1_zero<G>(ei);     // left neighbour's element, 0 for the first
                                if (G == 32) pdi = gl ? pdi : 0;
                                evict = min(evict, max(edi, dci));            // lane K-1: the evicted element, or a candidate that lost its race
                                ei = (edi > dci) ? ((pdi > dci) ? pidx : ic) : ei;        // larger elements move right
                                ed = __int_as_float(min(edi, max(pdi, dci)));
                            };
                            while (__any(gm != 0)) {                          // two steps per trip: halves the loop-carried register copies
                                step();
                                if (!__any(gm != 0)) break;
                                step();
                            }
                        } else {
                        mask_t gm = (__ballot(pass) >> (grp * G)) & (G == 64 ? ~0ull : ((1ull << G) - 1ull));
                        while (__any(gm != 0)) {                              // each group inserts its next passing candidate
                            const bool has = gm != 0;
                            const int l = has ? __builtin_ctzll(gm) : 0;
                            gm &= gm - 1;
                            float dc = __shfl(d2, l, G); int ic = __shfl(ci, l, G);
                            if (!has) { dc = INFINITY; ic = -1; }
                            const float pd = dpp_shr1_f<G>(ed); const int pidx = dpp_shr1_i<G>(ei);   // left neighbour's element
                            const bool gt = ed > dc || (LEX && ed == dc && (unsigned)ei > (unsigned)ic);      // larger elements move right
                            const bool left_gt = (gl > 0) && (pd > dc || (LEX && pd == dc && (unsigned)pidx > (unsigned)ic));
                            if (gl == K - 1) rejmin = fminf(rejmin, gt ? ed : dc);   // evicted element, or a candidate that lost its race
                            if (gt) { if (left_gt) { ed = pd; ei = pidx; } else { ed = dc; ei = ic; } }
                        }
                        }
                    }
                }
            }
        }
        // nothing outside the visited cube may enter the list or tie with its last entry (grid_core.h)
        const float bound2 = cbl_outside_bound2(g, uqx, uqy, uqz, cx, cy, cz, r);
        const float worst = __int_as_float(__builtin_amdgcn_ds_bpermute(last_src, __float_as_int(ed)));
        if (!done) done = (bound2 == INFINITY) || (worst < bound2);
        if (__all(done)) break;
    }

    // certify: list full, K distinct distances, no outside candidate tied with the K-th  (CblTopK::certify)
    const float worst = __int_as_float(__builtin_amdgcn_ds_bpermute(last_src, __float_as_int(ed)));
    const float rm = group_min_nonneg<G>(fminf(rejmin, gl == K - 1 ? __int_as_float(evict) : INFINITY));
    const float pd = dpp_shr1_f<G>(ed);
    const bool dup = (gl > 0) && (gl < K) && (ed == pd);
    const mask_t dm = (__ballot(dup) >> (grp * G)) & (G == 64 ? ~0ull : ((1ull << G) - 1ull));
    // set_exact: the caller only needs the reference's neighbour SET (sorted by distance); equal distances INSIDE the list
    // leave the set unambiguous, so only a tie at the K-th boundary (or an unfilled list) still needs the replay
    // set_exact == 2 ("any tie order"): a full list is final — among candidates tied at the K-th distance the visiting order decides
    // (set_exact == 1 keeps ONE piece of the reference's order: which of two equidistant nearest supports is column 0 — the CBL head drops
    //  "column 0 = the query itself", heads.py:195-196, and a coincident point would otherwise be free to take that place)
    const bool ok = (worst < INFINITY) && (set_exact == 2 || ((rm != worst) && (set_exact ? !(dm & 2ull) : dm == 0)));
    if (live) {
        if (gl < K) { idx[(size_t)q * K + gl] = ei; dist2[(size_t)q * K + gl] = ed; }
        if (!ok && gl == 0) worklist[atomicAdd(counters, 1)] = q;
    }
}

// ---- 3b. 16 < K <= 64: one WAVE per query, select-then-sort over the 27-cell block --------------------------------
// The insertion list above costs one serial step per accepted candidate (~K(1+ln(T/K)) of them).  Here the wave first loads
// ALL T candidates of the 3x3x3 block into registers (the 9 x-rows are one virtual range: all loads independent and coalesced),
// bisects a threshold tau with K <= #{d2 <= tau} <= 64 by ballot+popcount, compacts the survivors to one per lane through LDS
// and sorts them with a 21-stage bitonic network (DPP / ds_swizzle exchanges).  Lanes [0,K) then hold the ascending list, all
// other candidates feed `rejmin`, and the usual certification applies; further shells (rare) use the insertion step.

template <bool SELF, bool LEX>
__global__ __launch_bounds__(256) void knn_grid_wave_kernel(int b, int m, int K, const float* __restrict__ new_xyz,
                                                            const int* __restrict__ offset, const int* __restrict__ new_offset,
                                                            const CblGrid* __restrict__ grids, const int* __restrict__ cell_start,
                                                            const float4* __restrict__ sorted, int* __restrict__ idx, float* __restrict__ dist2,
                                                            int* __restrict__ worklist, int* __restrict__ counters, int set_exact,
                                                            int ks, int* __restrict__ idx_n, float* __restrict__ dist2_n,
                                                            int* __restrict__ worklist_n, int set_exact_n, CblCanon canon)
{
    __shared__ float2 slots[4][64];
    if ((int)blockIdx.x < canon.blocks) {                            // the exported cell order's canonical form (grid_order_canon_body), beside the queries
        grid_order_canon_body(canon.n, canon.pt_cell, cell_start, sorted, canon.order, blockIdx.x, canon.blocks);
        return;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int t = __builtin_amdgcn_readfirstlane(((int)blockIdx.x - canon.blocks) * 4 + wv);      // one query per wave
    if (t >= m) return;
    int q; float qx, qy, qz;
    if (SELF) { const float4 s = sorted[t]; q = __float_as_int(s.w); qx = s.x; qy = s.y; qz = s.z; }   // cell order
    else      { q = t; qx = new_xyz[3 * q]; qy = new_xyz[3 * q + 1]; qz = new_xyz[3 * q + 2]; }
    q = __builtin_amdgcn_readfirstlane(q);
    const int c = cbl_cloud_of(q, SELF ? offset : new_offset, b);
    const CblGrid g = grids[c];
    const float uqx = cbl_u(qx, g.ox, g.inv_cs), uqy = cbl_u(qy, g.oy, g.inv_cs), uqz = cbl_u(qz, g.oz, g.inv_cs);
    const int cx = cbl_cell_coord(uqx, g.nx), cy = cbl_cell_coord(uqy, g.ny), cz = cbl_cell_coord(uqz, g.nz);

    // ---- the 9 x-rows of the 27-cell block: lane r < 9 fetches row r's support range
    int rs = 0, rlen = 0;
    if (lane < 9) {
        const int y = cy + lane % 3 - 1, z = cz + lane / 3 - 1;
        if (y >= 0 && y < g.ny && z >= 0 && z < g.nz) {
            const int row = g.cell_base + g.nx * (y + g.ny * z);
            rs = cell_start[row + max(cx - 1, 0)];
            rlen = cell_start[row + min(cx + 1, g.nx - 1) + 1] - rs;
        }
    }
    int pre[9], delta[9], T = 0;                                     // wave-uniform: row r covers virtual positions [pre[r], pre[r]+len)
#pragma unroll
    for (int r = 0; r < 9; r++) {
        const int s_r = __builtin_amdgcn_readlane(rs, r), l_r = __builtin_amdgcn_readlane(rlen, r);
        pre[r] = T; delta[r] = s_r - T; T += l_r;
    }

    float ed = INFINITY; int ei = -1;                                // element `lane` of the ascending top list
    float rejmin = INFINITY;
    bool fast = (T >= K) && (T <= 64 * WV_NB);
    if (fast) {
        float d[WV_NB]; int id[WV_NB];
        // the row offsets once in vector registers: a select between a scalar and a vector under a lane mask needs two scalar
        // operands (gfx9 allows one), so the compiler would copy the scalar into a register at every use (3 VALU per row and block, not 2)
        int delta_v[9];
#pragma unroll
        for (int r = 0; r < 9; r++) delta_v[r] = kw_in_vgpr(delta[r]);
        // four blocks of 64 candidates at a time: their loads are all issued before the first distance is computed (one memory
        // round trip per four blocks instead of one per block — the wave's lifetime was dominated by those waits); positions
        // past T read the last candidate again (no divergent branch around the load) and are masked afterwards
#pragma unroll
        for (int j0 = 0; j0 < WV_NB; j0 += 4) {
            float4 p[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if ((j0 + u) * 64 < T) {
                    const int v = min((j0 + u) * 64 + lane, T - 1);
                    // row of position v.  (Resolving the rows on the scalar unit with real branches was measured: VALU -24 %, but the
                    // scalar unit — one per CU, as many issue slots as the four SIMDs together — became the bound: 61 -> 76 us.)
                    int dl = delta_v[0];
#pragma unroll
                    for (int r = 1; r < 9; r++) dl = (v >= pre[r]) ? delta_v[r] : dl;
                    p[u] = sorted[v + dl];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = j0 + u;
                d[j] = INFINITY; id[j] = -1;
                if (j * 64 < T) {
                    const bool in = j * 64 + lane < T;
                    const float dd = cbl_dist2(qx, qy, qz, p[u].x, p[u].y, p[u].z);      // knnquery_cuda_kernel.cu:99
                    d[j] = in ? dd : INFINITY; id[j] = in ? __float_as_int(p[u].w) : -1;
                }
            }
        }
        // ---- threshold: K <= #{d <= tau} <= 64
        float tau = 3.0e38f;
        if (T > 64) {
            // largest finite candidate distance, on the bit patterns (values >= +0 order like integers; only the last block is ragged)
            int mxi = 0;
#pragma unroll
            for (int j = 0; j < WV_NB; j++) {
                if ((j + 1) * 64 <= T) mxi = max(mxi, __float_as_int(d[j]));
                else if (j * 64 < T)   mxi = max(mxi, d[j] < INFINITY ? __float_as_int(d[j]) : 0);
            }
            mxi = max(mxi, dppx_i<0xB1>(mxi)); mxi = max(mxi, dppx_i<0x4E>(mxi));
            mxi = max(mxi, dppx_i<0x141>(mxi)); mxi = max(mxi, dppx_i<0x140>(mxi));
            float lo = -1.f, hi = __int_as_float(max(max(__builtin_amdgcn_readlane(mxi, 0), __builtin_amdgcn_readlane(mxi, 16)),
                                                     max(__builtin_amdgcn_readlane(mxi, 32), __builtin_amdgcn_readlane(mxi, 48))));
            const float first = (float)(K + 64) * 0.6f * __builtin_amdgcn_rcpf((float)T);   // a guess, not part of the result
            fast = false;
            for (int it = 0; it < 40; it++) {
                const float mid = it == 0 ? hi * first : lo + (hi - lo) * 0.5f;
                if (!(mid > lo && mid < hi)) break;                  // no float left between the brackets: ties, use the insertion path
                int cnt = 0;
#pragma unroll
                for (int j = 0; j < WV_NB; j++) if (j * 64 < T) cnt += __popcll(__ballot(d[j] <= mid));
                if (cnt < K) lo = mid; else if (cnt > 64) hi = mid; else { tau = mid; fast = true; break; }
            }
        }
        if (fast) {
            int base = 0;
#pragma unroll
            for (int j = 0; j < WV_NB; j++) {
                if (j * 64 < T) {
                    const bool pass = d[j] <= tau;
                    const unsigned long long mk = __ballot(pass);
                    if (pass) {
                        const int slot = base + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0));
                        slots[wv][slot] = make_float2(d[j], __int_as_float(id[j]));
                    } else rejmin = fminf(rejmin, d[j]);
                    base += __popcll(mk);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float sd = INFINITY; int si = -1;
            if (lane < base) { const float2 v = slots[wv][lane]; sd = v.x; si = __float_as_int(v.y); }
            wave_sort64<LEX>(sd, si, lane);
            if (lane < K) { ed = sd; ei = si; } else rejmin = fminf(rejmin, sd);
        }
    }

    bool done = false;
    int r = 1;
    if (fast) {
        const float bound2 = cbl_outside_bound2(g, uqx, uqy, uqz, cx, cy, cz, 1);
        done = (bound2 == INFINITY) || (rl_f(ed, K - 1) < bound2);
        r = 2;
    }
    for (; !done; r++) {
        // r == 1: the whole 3x3x3 block as 9 full rows (only when the fast path bailed out); r >= 2: shell r
        const int side = 2 * r + 1;
        for (int ri = 0; ri < side * side; ri++) {
            const int dz = ri / side - r, dy = ri % side - r;
            const bool full = (r == 1) || dz == -r || dz == r || dy == -r || dy == r;
            const int nseg = full ? 1 : 2;
            for (int sg = 0; sg < nseg; sg++) {
                int x0, x1;
                if (full) { x0 = max(cx - r, 0); x1 = min(cx + r, g.nx - 1); }
                else      { x0 = x1 = sg ? cx + r : cx - r; }
                const int y = cy + dy, z = cz + dz;
                int s = 0, e = 0;
                if (y >= 0 && y < g.ny && z >= 0 && z < g.nz && x0 >= 0 && x1 <= g.nx - 1) {
                    const int row = g.cell_base + g.nx * (y + g.ny * z);
                    s = cell_start[row + x0]; e = cell_start[row + x1 + 1];
                }
                s = __builtin_amdgcn_readfirstlane(s); e = __builtin_amdgcn_readfirstlane(e);
                for (int p = s; p < e; p += 64) {
                    const int pi = p + lane;
                    float d2 = INFINITY; int ci = -1;
                    if (pi < e) {
                        const float4 v = sorted[pi];
                        d2 = cbl_dist2(qx, qy, qz, v.x, v.y, v.z);
                        ci = __float_as_int(v.w);
                    }
                    const float worst = rl_f(ed, K - 1); const unsigned worst_i = LEX ? (unsigned)__builtin_amdgcn_readlane(ei, K - 1) : 0u;
                    const bool pass = d2 < worst || (LEX && d2 == worst && (unsigned)ci < worst_i);
                    rejmin = fminf(rejmin, pass ? INFINITY : d2);
                    unsigned long long gm = __ballot(pass);
                    while (gm) {
                        const int l = __builtin_ctzll(gm);
                        gm &= gm - 1;
                        const float dc = rl_f(d2, l); const int ic = __builtin_amdgcn_readlane(ci, l);
                        const float pd = dpp_shr1_f<64>(ed); const int pidx = dpp_shr1_i<64>(ei);
                        const bool gt = ed > dc || (LEX && ed == dc && (unsigned)ei > (unsigned)ic);
                        const bool left_gt = (lane > 0) && (pd > dc || (LEX && pd == dc && (unsigned)pidx > (unsigned)ic));
                        if (lane == K - 1) rejmin = fminf(rejmin, gt ? ed : dc);
                        if (gt) { if (left_gt) { ed = pd; ei = pidx; } else { ed = dc; ei = ic; } }
                    }
                }
            }
        }
        const float bound2 = cbl_outside_bound2(g, uqx, uqy, uqz, cx, cy, cz, r);
        done = (bound2 == INFINITY) || (rl_f(ed, K - 1) < bound2);
    }

    // certify (as in the group kernel)
    const float worst = rl_f(ed, K - 1);
    const float rm = group_min_nonneg<64>(rejmin);
    const float pd = dpp_shr1_f<64>(ed);
    const bool dup = (lane > 0) && (lane < K) && (ed == pd);
    const bool ok = (worst < INFINITY) && (set_exact == 2 || ((rm != worst) && (set_exact ? !(__ballot(dup) & 2ull) : __ballot(dup) == 0)));   // set: a tie for column 0 still replays
    if (lane < K) { idx[(size_t)q * K + lane] = ei; dist2[(size_t)q * K + lane] = ed; }
    if (!ok && lane == 0) worklist[atomicAdd(counters, 1)] = q;
    // the ks < K nearest as a result of their own (cbl_knnquery_nested): the list's distances are final whether or not its tie order
    // is, so the first ks entries stand unless THEY are decided by a tie — equal neighbours among the first ks (reference order only),
    // entry ks-1 tied with entry ks, or fewer than ks supports (the reference pads with 1e10, knnquery_cuda_kernel.cu:91-94)
    if (idx_n) {
        if (lane < ks) { idx_n[(size_t)q * ks + lane] = ei; dist2_n[(size_t)q * ks + lane] = ed; }
        const bool bad = (lane > 0 && lane <= ks && ed == pd && (lane == ks || lane == 1 || !set_exact_n)) || (lane == ks - 1 && !(ed < 1e10f));
        if (__ballot(bad) != 0ull && lane == 0) worklist_n[atomicAdd(counters + 1, 1)] = q;
    }
}

template <int G>
void launch_query(bool self, int b, int m, int K, const float* new_xyz, const int* offset, const int* new_offset, const Workspace& w,
                  int* idx, float* dist2, int set_exact, hipStream_t st)
{
    const long long waves = ((long long)m + (64 / G) - 1) / (64 / G);
    const dim3 grid(cbl_div_up(waves, 4)), block(256);
#define CBL_LAUNCH_GROUP(SELF_, LEX_) hipLaunchKernelGGL((knn_grid_group_kernel<G, SELF_, LEX_>), grid, block, 0, st, b, m, K, new_xyz, offset, new_offset, w.grids, \
                                                         w.cell_start, w.sorted, idx, dist2, w.worklist, w.counters, set_exact)
    if (set_exact == 2) { if (self) CBL_LAUNCH_GROUP(true, true); else CBL_LAUNCH_GROUP(false, true); }
    else                { if (self) CBL_LAUNCH_GROUP(true, false); else CBL_LAUNCH_GROUP(false, false); }
#undef CBL_LAUNCH_GROUP
}


// ---- N2: sorted radius neighbours, cropped to `limit` and padded with Ns -------------------------------------------
// Replaces batch_nanoflann_neighbors (tensorflow/ops/tf_custom_ops/tf_neighbors/neighbors/neighbors.cpp:213-336) plus the
// callers' crop to neighborhood_limits (tensorflow/datasets/base.py:756-765).  A group of G >= limit lanes per query; one pass over the
// 27-cell block suffices because the grid's cell edge is >= radius (cbl_grid_choose_radius).  counts[q] = true number of supports inside
// the ball (the reference's row length before padding).
// COLLECT, THEN RANK (round 4).  Round 3 kept the `limit` nearest in an ordered list across the lanes and inserted every in-ball candidate
// with a cross-lane shift (a serial chain of ~20 vector instructions and two LDS-crossbar reads per candidate: 0.84 of the vector issue
// slots).  Now the sweep only COLLECTS the supports with d2 < r^2 (strict, nanoflann.hpp:249-253) — their 64-bit keys (d2 bits, index) are
// compacted into a list of the group in LDS, up to 2 G of them — and the order is established afterwards by a rank count: a lane holds up
// to two of the keys and counts, over one broadcast read per list entry, how many keys are smaller; the rank IS the output column, so the
// result is stored straight from where it sits (no sort network, no permutation).  d2 >= +0, so the keys order like (d2, index): the
// canonical order of the oracle.  A ball with more than 2 G supports (never seen on the bench scenes) is selected by repeated minimum.
template <int G>
__device__ __forceinline__ unsigned long long group_min_u64(unsigned long long v)
{
#pragma unroll
    for (int s = G / 2; s >= 1; s >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, s, G), hi = __shfl_xor((unsigned)(v >> 32), s, G);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o < v ? o : v;
    }
    return v;
}

// position t of a query's virtual candidate range -> its key: rows start at c1 .. c8 (row 0 at 0), row r adds otab[r] (a table of the group in LDS:
// the row index is counted, the offset read).  The starts travel BY VALUE: with the whole table in a struct or captured by a lambda, LLVM turned a select
// chain over it into an indexed load from scratch memory (measured 3.8x slower); a select chain over 17 by-value registers measured 5 % slower than this.
__device__ __forceinline__ unsigned long long radius_key_lds(int t, int total, int c1, int c2, int c3, int c4, int c5, int c6, int c7, int c8, const int* __restrict__ otab,
                                                              float qx, float qy, float qz, float r2, const float4* __restrict__ sorted)
{
    const bool have = t < total;
    const int r = (t >= c1) + (t >= c2) + (t >= c3) + (t >= c4) + (t >= c5) + (t >= c6) + (t >= c7) + (t >= c8);
    const int o = otab[r];
    const float4 v = sorted[have ? t + o : 0];
    const float d2 = cbl_dist2(qx, qy, qz, v.x, v.y, v.z);
    return (have && d2 < r2) ? (((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)__float_as_int(v.w)) : ~0ull;
}

// U queries per group and kernel instance, level by level: every level of the dependent chain (query coordinates -> grid parameters -> the nine row
// bounds -> the candidates) is requested for all U queries before any of them is waited for.  The kernel is bound by that chain (0.57 of the wave
// cycles parked on memory at full occupancy, the vector ALU at 0.5): U = 2 halves the round trips per query.
template <int G, int U>
__global__ __launch_bounds__(256) void radius_group_kernel(int b, int nq, int ns_total, int limit, float r2, const float* __restrict__ queries,
                                                           const int* __restrict__ q_offset, const CblGrid* __restrict__ grids,
                                                           const int* __restrict__ cell_start, const float4* __restrict__ sorted,
                                                           int* __restrict__ out, int* __restrict__ counts, int* __restrict__ max_count)
{
    constexpr int QPW = 64 / G, CAP = 2 * G, NBF = G >= 64 ? 2 : 4;  // NBF batches of a query's range are requested at once
    using mask_t = unsigned long long;
    using key_t = unsigned long long;
    constexpr key_t NONE = ~0ull;
    __shared__ key_t slots[4 * QPW][CAP];
    __shared__ int otabs[4 * QPW][U][12];
    const int lane = threadIdx.x & 63, gl = lane & (G - 1), grp = lane / G;
    key_t* S = slots[(threadIdx.x >> 6) * QPW + grp];
    const int wave_global = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const mask_t below = (1ull << gl) - 1ull, gmask = G == 64 ? ~0ull : ((1ull << G) - 1ull);
    int q[U]; bool live[U]; float qx[U], qy[U], qz[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int t = (wave_global * QPW + grp) * U + u;
        live[u] = t < nq; q[u] = live[u] ? t : nq - 1;
        qx[u] = queries[3 * q[u]]; qy[u] = queries[3 * q[u] + 1]; qz[u] = queries[3 * q[u] + 2];
    }
    // The block's candidates as ONE virtual range: row r = 3 (dz + 1) + (dy + 1) is a contiguous range [s_r, e_r) of the cell-sorted supports;
    // lane r of the group fetches its row's two bounds (nine independent pairs of loads instead of nine dependent rounds), the lengths are
    // prefix-summed across the lanes, and position t of the concatenation maps to support t + off_r with r the last row whose start is <= t.
    int rs[U], re[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int c = cbl_cloud_of(q[u], q_offset, b);
        const CblGrid g = grids[c];
        const int cx = cbl_cell_coord(cbl_u(qx[u], g.ox, g.inv_cs), g.nx), cy = cbl_cell_coord(cbl_u(qy[u], g.oy, g.inv_cs), g.ny),
                  cz = cbl_cell_coord(cbl_u(qz[u], g.oz, g.inv_cs), g.nz);
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
        rs[u] = 0; re[u] = 0;
        if (gl < 9) {
            const int y = cy + (gl % 3) - 1, z = cz + (gl / 3) - 1;
            if (live[u] && g.end > g.start && y >= 0 && y < g.ny && z >= 0 && z < g.nz) {
                const int row = g.cell_base + g.nx * (y + g.ny * z);
                rs[u] = cell_start[row + x0]; re[u] = cell_start[row + x1 + 1];
            }
        }
    }
    // starts of the rows in the virtual range as plain locals, the offsets in a table of the group in LDS
    int c1[U], c2[U], c3[U], c4[U], c5[U], c6[U], c7[U], c8[U], total[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        int* otab = otabs[(threadIdx.x >> 6) * QPW + grp][u];
        int acc = 0;
#define CBL_ROW(r, cnext) { const int sr = __shfl(rs[u], r, G), er = __shfl(re[u], r, G); if (gl == 0) otab[r] = sr - acc; acc += er - sr; cnext = acc; }
        CBL_ROW(0, c1[u]) CBL_ROW(1, c2[u]) CBL_ROW(2, c3[u]) CBL_ROW(3, c4[u]) CBL_ROW(4, c5[u]) CBL_ROW(5, c6[u]) CBL_ROW(6, c7[u]) CBL_ROW(7, c8[u])
        CBL_ROW(8, total[u])                                         // total: group-uniform
#undef CBL_ROW
    }
#define CBL_KEY(u, t) radius_key_lds((t), total[u], c1[u], c2[u], c3[u], c4[u], c5[u], c6[u], c7[u], c8[u], otabs[(threadIdx.x >> 6) * QPW + grp][u], qx[u], qy[u], qz[u], r2, sorted)
    key_t first[U][NBF];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int k = 0; k < NBF; k++) first[u][k] = CBL_KEY(u, k * G + gl);
#pragma unroll
    for (int u = 0; u < U; u++) {
        int inside = 0;
        auto collect = [&](key_t key) __attribute__((always_inline)) {
            const bool in_ball = key != NONE;
            const mask_t gm = (__ballot(in_ball) >> (grp * G)) & gmask;
            const int slot = inside + __popcll(gm & below);
            if (in_ball && slot < CAP) S[slot] = key;
            inside += __popcll(gm);
        };
#pragma unroll
        for (int k = 0; k < NBF; k++) collect(first[u][k]);
        for (int t0 = NBF * G; __any(t0 < total[u]); t0 += G) collect(CBL_KEY(u, t0 + gl));
        const bool over = inside > CAP;                              // group-uniform
        if (!over) {
            const int n = inside;
            const key_t ka = gl < n ? S[gl] : NONE, kb = gl + G < n ? S[gl + G] : NONE;
            int ra = 0, rb = 0;
            for (int k = 0; __any(k < n); k += 4) {                 // four broadcast reads in flight per trip (the unused tail of the list holds stale keys: masked)
                key_t kk[4];
#pragma unroll
                for (int j = 0; j < 4; j++) kk[j] = S[min(k + j, CAP - 1)];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const bool on = k + j < n;
                    ra += (on && kk[j] < ka) ? 1 : 0; rb += (on && kk[j] < kb) ? 1 : 0;
                }
            }
            if (live[u]) {
                if (ka != NONE && ra < limit) out[(size_t)q[u] * limit + ra] = (int)(unsigned)ka;
                if (kb != NONE && rb < limit) out[(size_t)q[u] * limit + rb] = (int)(unsigned)kb;
                if (gl >= n && gl < limit) out[(size_t)q[u] * limit + gl] = ns_total;   // pad with supports.size(), neighbors.cpp:328 (limit <= G)
            }
        }
        if (__any(over)) {
            // more supports in the ball than the list holds: column p = the smallest key above column p - 1's, one sweep per column
            key_t prev = 0; bool none_yet = true;
            for (int col = 0; col < limit; col++) {
                key_t best = NONE;
                for (int t0 = 0; __any(over && t0 < total[u]); t0 += G) {
                    const key_t key = CBL_KEY(u, t0 + gl);
                    if ((none_yet || key > prev) && key < best) best = key;
                }
                best = group_min_u64<G>(best);
                if (over && live[u] && gl == 0) out[(size_t)q[u] * limit + col] = best == NONE ? ns_total : (int)(unsigned)best;
                prev = best; none_yet = false;
            }
        }
        if (live[u] && gl == 0) {
            if (counts) counts[q[u]] = inside;
            // one address for the whole launch: only groups that would raise the maximum touch it (monotone, so a stale read only costs a redundant atomic)
            if (inside > __hip_atomic_load(max_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(max_count, inside);
        }
    }
#undef CBL_KEY
}

}  // namespace

// build the per-cloud grids + cell-sorted supports into the workspace (shared with the radius search)
// per-cloud bounding boxes as order-preserving keys (bbox pre-set to 0xffffffff / 0 by the caller); shared with tfops.hip
int cbl_bbox_keys_launch(int b, int n, const float* xyz, const int* offset, unsigned* bbox, hipStream_t st)
{
    hipLaunchKernelGGL(grid_bbox_kernel, dim3(cbl_grid_for(n, 1024, 512)), dim3(256), 0, st, b, n, xyz, offset, bbox);
    return cbl_status();
}

__global__ __launch_bounds__(256) void grid_init_kernel(int b, int total, int* __restrict__ counters, unsigned* __restrict__ bbox, int* __restrict__ cell_count)
{
    const int i0 = blockIdx.x * 256 + threadIdx.x;
    if (i0 < 64) counters[i0] = 0;
    if (i0 < 6 * b) bbox[i0] = ((i0 % 6) < 3) ? 0xffffffffu : 0u;
    for (int i = i0; i < total; i += gridDim.x * 256) cell_count[i] = 0;
}

int cbl_grid_build(int b, int n, float pts_per_cell, const float* xyz, const int* offset, void* ws, hipStream_t st, int* order_out = nullptr, bool canon_later = false)
{
    // 5 launches: init+zero | bbox | grid params + histogram | tile scans | scan finish + scatter (+ the exported order's canonical form: a launch here, or —
    // canon_later — workgroups of the search launch that follows)
    Workspace w = carve(ws, b, n, 0);
    const int total = w.ncap + 1, ntiles = (total + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(grid_init_kernel, dim3(cbl_grid_for(total, 256, 512)), dim3(256), 0, st, b, total, w.counters, w.bbox, w.cell_count);
    hipLaunchKernelGGL(grid_bbox_kernel, dim3(cbl_grid_for(n, 1024, 512)), dim3(256), 0, st, b, n, xyz, offset, w.bbox);
    hipLaunchKernelGGL(grid_count_kernel, dim3(cbl_grid_for(n, 256)), dim3(256), 0, st, b, n, pts_per_cell, xyz, offset, w.bbox, w.grids, w.pt_cell, w.cell_count);
    hipLaunchKernelGGL(grid_scan_tiles_kernel, dim3(ntiles), dim3(256), 0, st, total, w.cell_count, w.cell_local, w.tile_sum);
    hipLaunchKernelGGL(grid_scatter_kernel, dim3(cbl_grid_for(n, 256)), dim3(256), 2 * sizeof(int) * (size_t)ntiles, st, n, total, ntiles, xyz, w.pt_cell,
                       w.tile_sum, w.cell_local, w.cell_start, w.cell_count, w.sorted, order_out);
    if (order_out && !canon_later)
        hipLaunchKernelGGL(grid_order_canon_kernel, dim3(cbl_grid_for(n, 256)), dim3(256), 0, st, n, w.pt_cell, w.cell_start, w.sorted, order_out);
    return cbl_status();
}

// ---- measurement support: how many candidate supports a self-search's 27-cell block holds per query (the pairs the wave / group kernels evaluate in their
// first round: what SURVEY 8(d) asks the cell-list search to report as "pairs visited") — recomputed from the grid a finished search left in its workspace
__global__ __launch_bounds__(256) void knn_block_candidates_kernel(int b, int n, const int* __restrict__ offset, const CblGrid* __restrict__ grids,
                                                                   const int* __restrict__ cell_start, const float4* __restrict__ sorted, int* __restrict__ count)
{
    const int t = blockIdx.x * 256 + threadIdx.x;                    // query = t-th support in cell order (the kernels' own enumeration of a self-search)
    if (t >= n) return;
    const float4 s = sorted[t];
    const int q = __float_as_int(s.w);
    const CblGrid g = grids[cbl_cloud_of(q, offset, b)];
    const int cx = cbl_cell_coord(cbl_u(s.x, g.ox, g.inv_cs), g.nx), cy = cbl_cell_coord(cbl_u(s.y, g.oy, g.inv_cs), g.ny), cz = cbl_cell_coord(cbl_u(s.z, g.oz, g.inv_cs), g.nz);
    int T = 0;
    for (int r = 0; r < 9; r++) {
        const int y = cy + r % 3 - 1, z = cz + r / 3 - 1;
        if (y >= 0 && y < g.ny && z >= 0 && z < g.nz) {
            const int row = g.cell_base + g.nx * (y + g.ny * z);
            T += cell_start[row + min(cx + 1, g.nx - 1) + 1] - cell_start[row + max(cx - 1, 0)];
        }
    }
    count[q] = T;
}

int cbl_knn_block_candidates_launch(int b, int n, const int* offset, void* ws, int* count, hipStream_t st)
{
    Workspace w = carve(ws, b, n, n);
    hipLaunchKernelGGL(knn_block_candidates_kernel, dim3(cbl_grid_for(n, 256, 1 << 20)), dim3(256), 0, st, b, n, offset, w.grids, w.cell_start, w.sorted, count);
    return cbl_status();
}

size_t cbl_knn_grid_workspace_bytes(int b, int n, int m, int nsample)
{
    // the grid pays off once the brute-force scan is long; tiny problems and huge K stay on the exact kernel
    if (nsample > GRID_MAX_K || n < 2048 || b <= 0) return 0;
    if ((long long)CELLS_PER_POINT * n + (long long)CELLS_PER_CLOUD * b > 0x3fffffffLL) return 0;
    return carve(nullptr, b, n, m).bytes;
}

// scratch of a finished grid search for a follow-up pass over its results (cbl_knnquery_nested): the worklist array and a counter that
// grid_init_kernel zeroed and the search did not touch; worklist2 / the same counter hold the narrow result's tied rows where the wave
// kernel already emitted it (CblKnnNarrow::fused)
void cbl_knn_grid_scratch(void* ws, int b, int n, int m, int** worklist, int** worklist2, int** zero_counter, const void** grids, const int** cell_start,
                          const void** sorted)
{
    Workspace w = carve(ws, b, n, m);
    *worklist = w.worklist; *worklist2 = w.worklist2; *zero_counter = w.counters + 1;
    *grids = w.grids; *cell_start = w.cell_start; *sorted = w.sorted;
}

int cbl_knn_grid_launch(int b, int n, int m, int nsample, const float* xyz, const float* new_xyz, const int* offset,
                        const int* new_offset, int* idx, float* dist2, void* ws, size_t ws_bytes, int set_exact, hipStream_t st, int* order_out,
                        CblKnnNarrow* narrow)
{
    Workspace w = carve(ws, b, n, m);
    if (ws_bytes < w.bytes) return CBL_ERR_WORKSPACE;
    // ~0.33*K points per cell if the cloud filled its bbox: the K-th neighbour is then usually inside the 27-cell block.  (0.42 until round 5; measured on the
    // bench scene, pairs visited / in-order step: 0.20: 7.4 M / 0.395 ms, 0.25: 9.1 M / 0.386, 0.33: 11.0 M / 0.388, 0.42: 13.0 M / 0.40, 0.60: 18.6 M / 0.416 —
    // smaller cells trade candidates for queries that need a second shell; CBL_KNN_CELL_FILL overrides for such sweeps)
    static const float fill = [] { const char* e = getenv("CBL_KNN_CELL_FILL"); const float v = e ? (float)atof(e) : 0.f; return (v > 0.05f && v < 4.f) ? v : 0.33f; }();
    const bool wave_search = nsample > 16;
    int rc = cbl_grid_build(b, n, fill * (float)(nsample < 4 ? 4 : nsample), xyz, offset, ws, st, order_out, wave_search);
    if (rc) return rc;
    const bool self = (new_xyz == xyz) && (m == n);
    if (wave_search) {                                               // select-then-sort, one wave per query
        CblCanon canon = {0, n, w.pt_cell, order_out};
        if (order_out) canon.blocks = (int)cbl_grid_for(n, 256);
        const dim3 grid(cbl_div_up(m, 4) + canon.blocks), block(256);
        // a narrower result of the same search rides along (its worklist: worklist2, counted in counters[1])
        const int ks = narrow ? narrow->nsample : 0;
        int* idx_n = narrow ? narrow->idx : nullptr; float* dist2_n = narrow ? narrow->dist2 : nullptr;
        const int set_n = narrow ? narrow->set_exact : 0;
        if (narrow) narrow->fused = true;
#define CBL_LAUNCH_WAVE(SELF_, LEX_) hipLaunchKernelGGL((knn_grid_wave_kernel<SELF_, LEX_>), grid, block, 0, st, b, m, nsample, new_xyz, offset, new_offset, w.grids, \
                                                       w.cell_start, w.sorted, idx, dist2, w.worklist, w.counters, set_exact, ks, idx_n, dist2_n, w.worklist2, set_n, canon)
        if (set_exact == 2) { if (self) CBL_LAUNCH_WAVE(true, true); else CBL_LAUNCH_WAVE(false, true); }
        else                { if (self) CBL_LAUNCH_WAVE(true, false); else CBL_LAUNCH_WAVE(false, false); }
#undef CBL_LAUNCH_WAVE
    }
    else launch_query<16>(self, b, m, nsample, new_xyz, offset, new_offset, w, idx, dist2, set_exact, st);   // 4 queries per wave
    rc = cbl_status();
    if (rc) return rc;
    if (narrow && narrow->fused && narrow->defer_replay) return CBL_OK;      // the caller replays both results' tied rows in one launch
    // exact replay of everything that was not certified (device-side count, no host sync)
    return cbl_knn_exact_worklist(b, m, nsample, xyz, new_xyz, self ? offset : offset, self ? offset : new_offset, idx, dist2,
                                  w.worklist, w.counters, m, st, w.grids, w.cell_start, w.sorted);
}

// N2 entry: queries (nq,3) with cumulative q_offset (b), supports (ns,3) with cumulative s_offset (b).
size_t cbl_radius_workspace_bytes_impl(int b, int ns) { return (b > 0 && ns >= 0) ? carve(nullptr, b, ns, 0).bytes : 0; }

int cbl_radius_launch(int b, int nq, int ns, const float* queries, const float* supports, const int* q_offset, const int* s_offset,
                      float radius, int limit, int* out, int* counts, int* max_count, void* ws, size_t ws_bytes, hipStream_t st, bool grid_is_built)
{
    Workspace w = carve(ws, b, ns, 0);
    if (ws_bytes < w.bytes) return CBL_ERR_WORKSPACE;
    // grid_is_built: the workspace still holds the grid of exactly these supports, offsets and radius (an earlier search on this stream built it)
    int rc = grid_is_built ? CBL_OK : cbl_grid_build(b, ns, -radius, supports, s_offset, ws, st);
    if (rc) return rc;
    hipError_t e = hipMemsetAsync(max_count, 0, sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    const int G = limit <= 16 ? 16 : limit <= 32 ? 32 : 64;
    const long long waves = ((long long)nq + (64 / G) - 1) / (64 / G);
    const dim3 grid(cbl_div_up(waves, 4)), block(256);
    const float r2 = radius * radius;                               // neighbors.cpp:230
    // queries per group and kernel instance: 2 where that still leaves >= 8 rounds of resident waves (200 000 queries: 129 -> 112 us; at 90 000
    // queries the shorter wave list costs more than the overlapped round trips save: 71 -> 83 us)
    const int ru = (long long)nq / ((64 / G) * 2) >= 49152 ? 2 : 1;
    const long long waves_u = ((long long)nq + (64 / G) * ru - 1) / ((64 / G) * ru);
    const dim3 grid_u(cbl_div_up(waves_u, 4));
#define CBL_RAD(G_, U_) hipLaunchKernelGGL((radius_group_kernel<G_, U_>), grid_u, block, 0, st, b, nq, ns, limit, r2, queries, q_offset, w.grids, w.cell_start, w.sorted, out, counts, max_count)
    if (ru == 2) { if (G == 16) CBL_RAD(16, 2); else if (G == 32) CBL_RAD(32, 2); else CBL_RAD(64, 2); }
    else         { if (G == 16) CBL_RAD(16, 1); else if (G == 32) CBL_RAD(32, 1); else CBL_RAD(64, 1); }
#undef CBL_RAD
    return cbl_status();
}
