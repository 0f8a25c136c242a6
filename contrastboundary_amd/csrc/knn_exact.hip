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

template <bool IN_LANES>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void knn_exact_wave_kernel(
    int b, int m, int K,
    const float* __restrict__ xyz, const float* __restrict__ new_xyz,
    const int* __restrict__ offset, const int* __restrict__ new_offset,
    int* __restrict__ idx, float* __restrict__ dist2,
    const int* __restrict__ worklist, const int* __restrict__ worklist_count)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = blockIdx.x * WAVES_PER_BLOCK + wave;              // wave-uniform work item

    // optional indirection: only the queries listed in worklist[0 .. *worklist_count)
    int q;
    if (worklist) {
        const int n_active = *worklist_count;
        if (w >= n_active) return;
        q = worklist[w];
    } else {
        if (w >= m) return;
        q = w;
    }
    q = __builtin_amdgcn_readfirstlane(q);

    const int c = cbl_cloud_of(q, new_offset, b);
    const int start = (c == 0) ? 0 : offset[c - 1];                  // :75-79
    const int end = offset[c];                                       // :80
    const float qx = new_xyz[3 * q + 0], qy = new_xyz[3 * q + 1], qz = new_xyz[3 * q + 2];

    typename std::conditional<IN_LANES, LaneHeap, LdsHeap>::type h;
    if constexpr (!IN_LANES) {
        h.d = reinterpret_cast<float*>(smem) + (size_t)wave * 2 * K;
        h.i = reinterpret_cast<int*>(h.d + K);
    }
    h.init(K, 1e10f, start);                                         // :91-94
    float root = 1e10f;

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
                    heap_replace_root(h, K, dl, base + 64 * u + l);
                    root = h.D(0);
                }
            }
        }
    }

    // heap_sort(), :39-48
    for (int last = K - 1; last > 0; last--) {
        const float d = h.D(last); const int id = h.I(last);
        h.set(last, h.D(0), h.I(0));
        heap_replace_root(h, last, d, id);
    }
    int* orow = idx + (size_t)q * K; float* drow = dist2 + (size_t)q * K;
    if constexpr (IN_LANES) {
        if (lane < K) { orow[lane] = h.hi; drow[lane] = h.hd; }
    } else {
        for (int j = lane; j < K; j += 64) { orow[j] = h.i[j]; drow[j] = h.d[j]; }
    }
}

}  // namespace

static int launch_knn_exact(int b, int m, int K, const float* xyz, const float* new_xyz, const int* offset,
                            const int* new_offset, int* idx, float* dist2,
                            const int* worklist, const int* worklist_count, int max_work, hipStream_t st)
{
    const int nq = worklist ? max_work : m;
    if (nq <= 0) return CBL_OK;
    const dim3 grid(cbl_div_up(nq, WAVES_PER_BLOCK)), block(64 * WAVES_PER_BLOCK);
    if (K <= 64)
        hipLaunchKernelGGL(knn_exact_wave_kernel<true>, grid, block, 0, st,
                           b, m, K, xyz, new_xyz, offset, new_offset, idx, dist2, worklist, worklist_count);
    else
        hipLaunchKernelGGL(knn_exact_wave_kernel<false>, grid, block, (size_t)WAVES_PER_BLOCK * K * 8, st,
                           b, m, K, xyz, new_xyz, offset, new_offset, idx, dist2, worklist, worklist_count);
    return cbl_status();
}

// used by knn_grid.hip for the exact replay of tied queries
int cbl_knn_exact_worklist(int b, int m, int K, const float* xyz, const float* new_xyz, const int* offset,
                           const int* new_offset, int* idx, float* dist2,
                           const int* worklist, const int* worklist_count, int max_work, hipStream_t st)
{
    return launch_knn_exact(b, m, K, xyz, new_xyz, offset, new_offset, idx, dist2, worklist, worklist_count, max_work, st);
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
