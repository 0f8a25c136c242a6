// K1 (exact): segmented brute-force KNN with the reference's heap semantics, bit for bit.
// Replaces knnquery_cuda_kernel  /root/reference/pytorch/lib/pointops/src/knnquery/knnquery_cuda_kernel.cu:65-111.
//
// MI355X mapping (not the reference's): one LANE per query, one WAVE per workgroup so that even
// m = 40960 queries give 640 independent workgroups across the 256 CUs.  The support index `i` is
// wave-uniform: its coordinates come through the scalar cache into SGPRs (no per-lane reloads of
// xyz as in the reference) and the only per-lane state in the hot loop is the query point, the
// [start,end) range of its cloud and the heap root.  The K-entry max-heap lives in LDS, k-major
// (slot*64 + lane) so that lanes touching the same slot never bank-conflict; for nsample too large
// for LDS the heap lives in the caller's idx/dist2 rows themselves.
//
// The visiting order is the reference's (ascending support index), every comparison keeps its
// strictness (d2 < root; right child only if strictly larger; stop only if parent strictly larger),
// so ties resolve identically — see oracle/pointops_oracle.c for the CPU statement of the same.
#include "cbl_common.h"

namespace {

constexpr int KNN_BLOCK = 64;

// Heap storage accessors.  LDS: element j of this lane at base[j * 64]; global: base[j].
template <bool IN_LDS> struct HeapRef {
    float* d; int* i;
    __device__ __forceinline__ float& D(int j) const { return IN_LDS ? d[j * KNN_BLOCK] : d[j]; }
    __device__ __forceinline__ int&   I(int j) const { return IN_LDS ? i[j * KNN_BLOCK] : i[j]; }
};

// Put (d, id) at the root of a max-heap of `len` entries and sift it down ("hole" form of
// reheap(), knnquery_cuda_kernel.cu:21-36: identical final layout, fewer stores).
template <bool IN_LDS>
__device__ __forceinline__ void heap_replace_root(const HeapRef<IN_LDS>& h, int len, float d, int id)
{
    int parent = 0;
    for (;;) {
        int kid = 2 * parent + 1;
        if (kid >= len) break;
        float kd = h.D(kid);
        if (kid + 1 < len) {
            const float rd = h.D(kid + 1);
            if (rd > kd) { kd = rd; kid += 1; }          // right child only when strictly larger
        }
        if (d > kd) break;                                // stop only when strictly larger
        h.D(parent) = kd; h.I(parent) = h.I(kid);
        parent = kid;
    }
    h.D(parent) = d; h.I(parent) = id;
}

template <bool IN_LDS>
__global__ __launch_bounds__(KNN_BLOCK) void knn_exact_kernel(
    int b, int m, int K,
    const float* __restrict__ xyz, const float* __restrict__ new_xyz,
    const int* __restrict__ offset, const int* __restrict__ new_offset,
    int* __restrict__ idx, float* __restrict__ dist2,
    const int* __restrict__ worklist, const int* __restrict__ worklist_count)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;

    // optional indirection: only the queries listed in worklist[0 .. *worklist_count) (tied queries
    // handed over by the grid kernel); otherwise queries are blockIdx.x*64 + lane.
    int q, n_active;
    if (worklist) {
        n_active = *worklist_count;
        const int w = blockIdx.x * KNN_BLOCK + lane;
        if (blockIdx.x * KNN_BLOCK >= n_active) return;
        q = (w < n_active) ? worklist[w] : -1;
    } else {
        n_active = m;
        q = blockIdx.x * KNN_BLOCK + lane;
        if (q >= m) q = -1;
    }
    const bool live = q >= 0;

    int start = 0, end = 0;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (live) {
        const int c = cbl_cloud_of(q, new_offset, b);
        start = (c == 0) ? 0 : offset[c - 1];
        end = offset[c];
        qx = new_xyz[3 * q + 0]; qy = new_xyz[3 * q + 1]; qz = new_xyz[3 * q + 2];
    }
    // wave-uniform union of the lanes' support ranges
    int lo = live ? start : 0x7fffffff, hi = live ? end : 0;
    for (int s = 32; s >= 1; s >>= 1) {
        lo = min(lo, __shfl_xor(lo, s));
        hi = max(hi, __shfl_xor(hi, s));
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);

    HeapRef<IN_LDS> h;
    if (IN_LDS) {
        h.d = reinterpret_cast<float*>(smem) + lane;
        h.i = reinterpret_cast<int*>(smem) + K * KNN_BLOCK + lane;
    } else {
        h.d = dist2 + (size_t)(live ? q : 0) * K;
        h.i = idx + (size_t)(live ? q : 0) * K;
    }
    if (IN_LDS || live)
        for (int j = 0; j < K; j++) { h.D(j) = 1e10f; h.I(j) = start; }     // :91-94

    float root = 1e10f;
    const unsigned span = (unsigned)(end - start);
    auto consider = [&](int i, float sx, float sy, float sz) {
        const float d2 = cbl_dist2(qx, qy, qz, sx, sy, sz);                  // (new - x)^2 ..., :99
        const bool mine = (unsigned)(i - start) < span;
        if (mine && d2 < root) {                                             // strict, :100
            heap_replace_root(h, K, d2, i);
            root = h.D(0);
        }
    };
    // wave-uniform addresses -> scalar loads; 8 supports are fetched ahead of their use so the
    // scalar-cache latency overlaps the (rare, divergent) heap updates
    constexpr int U = 8;
    int i = lo;
    for (; i + U <= hi; i += U) {
        float s[3 * U];
#pragma unroll
        for (int t = 0; t < 3 * U; t++) s[t] = xyz[3 * i + t];
#pragma unroll
        for (int t = 0; t < U; t++) consider(i + t, s[3 * t], s[3 * t + 1], s[3 * t + 2]);
    }
    for (; i < hi; i++) consider(i, xyz[3 * i + 0], xyz[3 * i + 1], xyz[3 * i + 2]);
    if (!live) return;

    // heap_sort(), :39-48
    for (int last = K - 1; last > 0; last--) {
        const float d = h.D(last); const int id = h.I(last);
        h.D(last) = h.D(0); h.I(last) = h.I(0);
        heap_replace_root(h, last, d, id);
    }
    if (IN_LDS) {
        int* orow = idx + (size_t)q * K; float* drow = dist2 + (size_t)q * K;
        for (int j = 0; j < K; j++) { orow[j] = h.I(j); drow[j] = h.D(j); }
    }
}

}  // namespace

// LDS heap while nsample*64*8 B fits in the CU's 160 KiB, else heap in the output rows.
static int launch_knn_exact(int b, int m, int K, const float* xyz, const float* new_xyz, const int* offset,
                            const int* new_offset, int* idx, float* dist2,
                            const int* worklist, const int* worklist_count, int max_work, hipStream_t st)
{
    const int nq = worklist ? max_work : m;
    if (nq <= 0) return CBL_OK;
    const unsigned grid = cbl_div_up(nq, KNN_BLOCK);
    const size_t lds = (size_t)K * KNN_BLOCK * 8;
    if (lds <= 160 * 1024) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_exact_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
        hipLaunchKernelGGL(knn_exact_kernel<true>, dim3(grid), dim3(KNN_BLOCK), lds, st,
                           b, m, K, xyz, new_xyz, offset, new_offset, idx, dist2, worklist, worklist_count);
    } else {
        hipLaunchKernelGGL(knn_exact_kernel<false>, dim3(grid), dim3(KNN_BLOCK), 0, st,
                           b, m, K, xyz, new_xyz, offset, new_offset, idx, dist2, worklist, worklist_count);
    }
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
