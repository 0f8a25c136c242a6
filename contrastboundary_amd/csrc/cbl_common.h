// Shared device/host helpers for libcbl_amd.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include "../../include/cbl_amd.h"

#define CBL_EXPORT extern "C" __attribute__((visibility("default")))
#define CBL_WAVE 64

static inline int cbl_status()
{
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? CBL_OK : (int)e;
}

static inline hipStream_t cbl_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline unsigned cbl_div_up(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

// grid size for a grid-stride elementwise kernel: enough blocks to fill 256 CUs x 8, never 0
static inline unsigned cbl_grid_for(long long work_items, int block, int max_blocks = 256 * 16)
{
    long long g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (unsigned)g;
}

// Workgroups of `kernel` that are resident at once on the current device (occupancy x compute units), for persistent launches: a kernel that walks its
// work with a grid-stride loop is launched with at most this many workgroups, so a wave takes several trips (what it fetches a trip ahead gets used) and
// the dispatcher is not kept busy with thousands of one-trip workgroups while other streams' kernels wait for slots.  Cached per (kernel, LDS, device).
// Launches come from several host threads at once (the caller's, autograd's backward thread, a pyramid loader thread): the table is guarded by a mutex
// (one shared instance across translation units: an inline function's statics).
inline unsigned cbl_resident_blocks(const void* kernel, int block, size_t dynamic_lds = 0)
{
    struct Entry { const void* fn; size_t lds; int dev; unsigned n; };
    static Entry cache[64] = {};
    static std::mutex guard;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    {
        std::lock_guard<std::mutex> hold(guard);
        for (int i = 0; i < 64 && cache[i].fn; i++)
            if (cache[i].fn == kernel && cache[i].lds == dynamic_lds && cache[i].dev == dev) return cache[i].n;
    }
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, dynamic_lds) != hipSuccess || per_cu <= 0) per_cu = 4;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const unsigned n = ((unsigned)(per_cu * cus) + 7u) & ~7u;           // a multiple of the 8 XCDs
    std::lock_guard<std::mutex> hold(guard);
    for (int i = 0; i < 64; i++) {
        if (cache[i].fn == kernel && cache[i].lds == dynamic_lds && cache[i].dev == dev) break;      // another thread was faster
        if (!cache[i].fn) { cache[i].lds = dynamic_lds; cache[i].dev = dev; cache[i].n = n; cache[i].fn = kernel; break; }
    }
    return n;
}
template <class F> static inline unsigned cbl_persistent_grid(unsigned wanted, F kernel, int block, size_t dynamic_lds = 0)
{
    const unsigned r = cbl_resident_blocks(reinterpret_cast<const void*>(kernel), block, dynamic_lds);
    return wanted < r ? wanted : r;
}

// cloud of a stacked row: first c with row < ends[c]  (knnquery_cuda_kernel.cu:51-62 does this by
// linear scan; binary search gives the same answer for non-decreasing ends, empty clouds included)
__device__ __forceinline__ int cbl_cloud_of(int row, const int* __restrict__ ends, int b)
{
    int lo = 0, hi = b - 1;          // answer in [0, b-1]; rows >= ends[b-1] clamp to b-1
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (row < ends[mid]) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// squared distance with the reference's association and no FMA contraction
// (knnquery_cuda_kernel.cu:99; the library is built with -ffp-contract=off)
__device__ __forceinline__ float cbl_dist2(float ax, float ay, float az, float bx, float by, float bz)
{
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return (dx * dx + dy * dy) + dz * dz;
}

__device__ __forceinline__ bool cbl_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline bool cbl_host_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// q = p / d for any 32-bit p and a divisor fixed per launch: umulhi(p, floor(2^32 / d)) is q or q - 1
struct CblFastDiv { unsigned d, m; };
static inline CblFastDiv cbl_fastdiv_make(unsigned d) { CblFastDiv f; f.d = d; f.m = d <= 1 ? 0xffffffffu : (unsigned)(0x100000000ull / d); return f; }
__device__ __forceinline__ unsigned cbl_fastdiv(unsigned p, CblFastDiv f)
{
    if (f.d == 1) return p;
    unsigned q = __umulhi(p, f.m);
    if (p - q * f.d >= f.d) q++;
    return q;
}

// ---- processing order ("*_ordered" entry points) ----------------------------------------------------------------------------
// The point-walking kernels take an optional `order`: the sequence in which the points are processed (a permutation; the cell order of
// the neighbour search, cbl_knnquery_ordered).  Workgroup b of a launch runs on XCD b % 8 (round-robin dispatch); a kernel walks
// VIRTUAL workgroups v = blockIdx.x, + gridDim.x, ... with gridDim.x a multiple of 8, and virtual workgroup v takes sequence slot
// cbl_xcd_slot(v, nwg): XCD x then works on the contiguous eighth [x * per, (x + 1) * per) of the sequence — one slab of the scene,
// whose rows stay in that XCD's 4 MB L2.  (On a device partitioned differently the mapping is merely another permutation.)
__device__ __forceinline__ unsigned cbl_xcd_per(unsigned nwg) { return (nwg + 7u) >> 3; }
__device__ __forceinline__ unsigned cbl_xcd_slot(unsigned v, unsigned nwg) { return (v & 7u) * cbl_xcd_per(nwg) + (v >> 3); }
static inline unsigned cbl_round_up8(unsigned g) { return (g + 7u) & ~7u; }


// A narrower result (the first `nsample` of the wider list) that a grid search may emit along with its own (cbl_knnquery_nested):
// filled in by the caller; `fused` comes back true where the search kernel wrote idx / dist2 and listed the rows decided by a tie
// (its second worklist, counters[1]) — otherwise the caller derives them in a pass of its own.
// `defer_replay` (in): leave the replay of the WIDE result's tied rows to the caller as well (it then runs both replays in one launch).
struct CblKnnNarrow { int nsample; int set_exact; int* idx; float* dist2; bool fused; bool defer_replay; };
