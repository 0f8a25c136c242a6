// K1 front door: picks the grid kernels (K <= 64, certified, knn_grid.hip), the block-select kernel (64 < K <= 1024, certified,
// knn_select.hip) or the brute-force kernel (small problems, knn_exact.hip).
#include "cbl_common.h"

size_t cbl_knn_grid_workspace_bytes(int b, int n, int m, int nsample);     // knn_grid.hip
int cbl_knn_grid_launch(int b, int n, int m, int nsample, const float* xyz, const float* new_xyz, const int* offset,
                        const int* new_offset, int* idx, float* dist2, void* ws, size_t ws_bytes, int set_exact, hipStream_t st, int* order_out, CblKnnNarrow* narrow);

size_t cbl_knn_select_workspace_bytes(int b, int n, int m, int nsample);   // knn_select.hip
int cbl_knn_select_launch(int b, int n, int m, int nsample, const float* xyz, const float* new_xyz, const int* offset, const int* new_offset,
                          int* idx, float* dist2, void* ws, size_t ws_bytes, int set_exact, hipStream_t st);

CBL_EXPORT size_t cbl_knnquery_workspace_bytes(int b, int n, int m, int nsample)
{
    const size_t g = cbl_knn_grid_workspace_bytes(b, n, m, nsample);
    return g ? g : cbl_knn_select_workspace_bytes(b, n, m, nsample);
}

static int knnquery_impl(int b, int n, int m, int nsample, const float* xyz, const float* new_xyz,
                         const int* offset, const int* new_offset, int* idx, float* dist2,
                         void* workspace, size_t workspace_bytes, int set_exact, void* stream, int* order_out = nullptr, CblKnnNarrow* narrow = nullptr)
{
    if (b <= 0 || n < 0 || m < 0 || nsample <= 0 || nsample > CBL_KNN_MAX_NSAMPLE) return CBL_ERR_BAD_ARG;
    if (m == 0) return CBL_OK;
    if (!xyz || !new_xyz || !offset || !new_offset || !idx || !dist2) return CBL_ERR_BAD_ARG;
    const size_t need = cbl_knn_grid_workspace_bytes(b, n, m, nsample);
    if (need > 0 && workspace && workspace_bytes >= need)
        return cbl_knn_grid_launch(b, n, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, workspace, workspace_bytes, set_exact, cbl_stream(stream), order_out, narrow);
    if (order_out) return CBL_ERR_UNSUPPORTED;                      // only the grid path sorts the supports into cells
    const size_t need_sel = cbl_knn_select_workspace_bytes(b, n, m, nsample);
    if (need_sel > 0 && workspace && workspace_bytes >= need_sel)
        return cbl_knn_select_launch(b, n, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, workspace, workspace_bytes, set_exact, cbl_stream(stream));
    return cbl_knnquery_exact(b, n, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, stream);
}

CBL_EXPORT int cbl_knnquery(int b, int n, int m, int nsample, const float* xyz, const float* new_xyz,
                            const int* offset, const int* new_offset, int* idx, float* dist2,
                            void* workspace, size_t workspace_bytes, void* stream)
{
    return knnquery_impl(b, n, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, workspace, workspace_bytes, 0, stream);
}

CBL_EXPORT int cbl_knnquery_set(int b, int n, int m, int nsample, const float* xyz, const float* new_xyz,
                                const int* offset, const int* new_offset, int* idx, float* dist2,
                                void* workspace, size_t workspace_bytes, void* stream)
{
    return knnquery_impl(b, n, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, workspace, workspace_bytes, 1, stream);
}

CBL_EXPORT int cbl_knnquery_anytie(int b, int n, int m, int nsample, const float* xyz, const float* new_xyz,
                                   const int* offset, const int* new_offset, int* idx, float* dist2,
                                   void* workspace, size_t workspace_bytes, void* stream)
{
    return knnquery_impl(b, n, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, workspace, workspace_bytes, 2, stream);
}

CBL_EXPORT int cbl_knnquery_ordered(int b, int n, int m, int nsample, const float* xyz, const float* new_xyz,
                                    const int* offset, const int* new_offset, int* idx, float* dist2, int tie_policy, int* cell_order,
                                    void* workspace, size_t workspace_bytes, void* stream)
{
    if (tie_policy < 0 || tie_policy > 2 || !cell_order) return CBL_ERR_BAD_ARG;
    return knnquery_impl(b, n, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, workspace, workspace_bytes, tie_policy, stream, cell_order);
}

int cbl_knn_block_candidates_launch(int b, int n, const int* offset, void* ws, int* count, hipStream_t st);    // knn_grid.hip

// measurement support (bench.py roofline.search): candidates per query of the grid search that last used `workspace` for a SELF-search over these (b, n)
CBL_EXPORT int cbl_knn_grid_block_candidates(int b, int n, int nsample, const int* offset, int* count, void* workspace, size_t workspace_bytes, void* stream)
{
    const size_t need = cbl_knn_grid_workspace_bytes(b, n, n, nsample);
    if (need == 0) return CBL_ERR_UNSUPPORTED;
    if (workspace_bytes < need) return CBL_ERR_WORKSPACE;
    return cbl_knn_block_candidates_launch(b, n, offset, workspace, count, cbl_stream(stream));
}

// ---- a narrower search from a wider one over the same (supports, queries) ----------------------------------------------------
// The K nearest neighbours are the first K of the K' > K nearest.  A row of the wider result (ascending distances, exact K' smallest —
// every tie policy delivers that) gives the narrower result directly unless a tie decides it: equal distances among its first K entries
// (the reference lists them in its heap's order: policy 0 only) or across the K / K+1 boundary (which of them belong to the reference's
// set: policies 0 and 1), or a row that is not full (fewer than K supports).  Those queries go to the exact replay, the same
// certification the grid kernels apply to their own lists.  One search instead of two for networks that look at one geometry with
// several neighbourhood sizes (the blocks' K = 8 / 16 and the CBL head's K = 36 at a stage).
int cbl_knn_exact_worklist(int b, int m, int K, const float* xyz, const float* new_xyz, const int* offset, const int* new_offset,
                           int* idx, float* dist2, const int* worklist, const int* worklist_count, int max_work, hipStream_t st,
                           const void* grids, const int* cell_start, const void* sorted);   // knn_exact.hip

namespace {
__global__ __launch_bounds__(256) void knn_prefix_kernel(int m, int kb, int ks, const int* __restrict__ idx_big, const float* __restrict__ d2_big,
                                                         int* __restrict__ idx, float* __restrict__ dist2, int set_exact,
                                                         int* __restrict__ worklist, int* __restrict__ counter)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= m) return;
    const int* ib = idx_big + (size_t)q * kb; const float* db = d2_big + (size_t)q * kb;
    float prev = -1.f; bool dup = false, dup0 = false;
    for (int j = 0; j < ks; j++) {
        const float d = db[j];
        idx[(size_t)q * ks + j] = ib[j]; dist2[(size_t)q * ks + j] = d;
        dup = dup || (d == prev);
        dup0 = dup0 || (j == 1 && d == prev);                      // a tie for column 0: replayed under the set policy too (knn_grid.hip)
        prev = d;
    }
    const bool full = prev < 1e10f;                                 // the reference's initial heap entries are 1e10 (knnquery_cuda_kernel.cu:91-94)
    const bool boundary = db[ks] == prev;
    const bool ok = full && !boundary && (set_exact ? !dup0 : !dup);
    if (!ok) worklist[atomicAdd(counter, 1)] = q;
}
// the same for K a power of two <= 64: K consecutive lanes per query (a row of the output is one coalesced segment), ties found by comparing
// with the next column, gathered per query with one ballot
template <int KS>
__global__ __launch_bounds__(256) void knn_prefix_pow2_kernel(int m, int kb, const int* __restrict__ idx_big, const float* __restrict__ d2_big,
                                                              int* __restrict__ idx, float* __restrict__ dist2, int set_exact,
                                                              int* __restrict__ worklist, int* __restrict__ counter)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const long long qq = e / KS;
    const int j = (int)(e - qq * KS);
    const bool live = qq < m;
    const int q = live ? (int)qq : m - 1;
    const float d = d2_big[(size_t)q * kb + j], dn = d2_big[(size_t)q * kb + j + 1];       // j + 1 <= KS < kb
    if (live) { idx[(size_t)q * KS + j] = idx_big[(size_t)q * kb + j]; dist2[(size_t)q * KS + j] = d; }
    const bool last = j == KS - 1;
    const bool bad = (d == dn && (last || j == 0 || !set_exact)) || (last && !(d < 1e10f));        // set policy: boundary ties and a tie for column 0
    const unsigned long long bm = __ballot(bad && live);
    const unsigned long long gm = (KS == 64) ? bm : ((bm >> (lane & ~(KS - 1))) & ((1ull << KS) - 1ull));
    if (live && j == 0 && gm != 0ull) worklist[atomicAdd(counter, 1)] = q;
}
}  // namespace

void cbl_knn_grid_scratch(void* ws, int b, int n, int m, int** worklist, int** worklist2, int** zero_counter, const void** grids,
                          const int** cell_start, const void** sorted);      // knn_grid.hip
int cbl_knn_exact_worklist2(int b, int m, const float* xyz, const float* new_xyz, const int* offset, const int* new_offset,
                            int K1, int* idx1, float* dist2_1, const int* worklist1, const int* count1,
                            int K2, int* idx2, float* dist2_2, const int* worklist2, const int* count2,
                            hipStream_t st, const void* grids, const int* cell_start, const void* sorted);     // knn_exact.hip

static int launch_prefix(int m, int nsample_wide, int nsample, const int* idx_wide, const float* dist2_wide, int* idx, float* dist2, int tie_policy,
                         int* worklist, int* counter, hipStream_t st)
{
#define CBL_PREFIX_POW2(KS) hipLaunchKernelGGL(knn_prefix_pow2_kernel<KS>, dim3(cbl_div_up((long long)m * KS, 256)), dim3(256), 0, st, m, nsample_wide, idx_wide, dist2_wide, \
                                               idx, dist2, tie_policy, worklist, counter)
    switch (nsample) {
        case 1: CBL_PREFIX_POW2(1); break;
        case 2: CBL_PREFIX_POW2(2); break;
        case 4: CBL_PREFIX_POW2(4); break;
        case 8: CBL_PREFIX_POW2(8); break;
        case 16: CBL_PREFIX_POW2(16); break;
        case 32: CBL_PREFIX_POW2(32); break;
        default:
            hipLaunchKernelGGL(knn_prefix_kernel, dim3(cbl_div_up(m, 256)), dim3(256), 0, st, m, nsample_wide, nsample, idx_wide, dist2_wide, idx, dist2, tie_policy, worklist, counter);
    }
#undef CBL_PREFIX_POW2
    return cbl_status();
}

// the wide search and one narrower search derived from it in ONE call: the derivation reuses the grid search's scratch (its worklist array
// and a counter the build zeroed), so no counter reset of its own is launched
CBL_EXPORT int cbl_knnquery_nested(int b, int n, int m, int nsample_wide, int tie_policy_wide, int nsample, int tie_policy,
                                   const float* xyz, const float* new_xyz, const int* offset, const int* new_offset,
                                   int* idx_wide, float* dist2_wide, int* idx, float* dist2, int* cell_order, void* event_after_wide,
                                   void* workspace, size_t workspace_bytes, void* stream)
{
    if (nsample <= 0 || nsample >= nsample_wide || tie_policy < 0 || tie_policy > 1 || tie_policy_wide < 0 || tie_policy_wide > 2 || !idx || !dist2) return CBL_ERR_BAD_ARG;
    const size_t need = cbl_knn_grid_workspace_bytes(b, n, m, nsample_wide);
    if (need == 0 || !workspace || workspace_bytes < need) return CBL_ERR_UNSUPPORTED;      // only behind the grid path (its scratch is what is reused)
    // with the wave kernel (nsample_wide > 16) the narrow rows and the list of those a tie decides come out of the search itself
    CblKnnNarrow narrow{nsample, tie_policy, idx, dist2, false, true};
    int rc = knnquery_impl(b, n, m, nsample_wide, xyz, new_xyz, offset, new_offset, idx_wide, dist2_wide, workspace, workspace_bytes, tie_policy_wide, stream, cell_order, &narrow);
    if (rc || m == 0) return rc;
    hipStream_t st = cbl_stream(stream);
    int *worklist, *worklist2, *counter; const void *grids, *sorted; const int* cell_start;
    cbl_knn_grid_scratch(workspace, b, n, m, &worklist, &worklist2, &counter, &grids, &cell_start, &sorted);
    if (narrow.fused) {
        // the wave kernel left both worklists behind (wide: worklist / counters[0], narrow: worklist2 / counters[1]) and skipped the wide
        // replay: one launch replays both, different workgroups each (a replay occupies one or two workgroups: they ran one after the other before)
        const bool self = (new_xyz == xyz) && (m == n);
        rc = cbl_knn_exact_worklist2(b, m, xyz, new_xyz, offset, self ? offset : new_offset, nsample_wide, idx_wide, dist2_wide, worklist, counter - 1,
                                     nsample, idx, dist2, worklist2, counter, st, grids, cell_start, sorted);
        if (rc) return rc;
        // consumers of the WIDE result on other streams wait for this
        if (event_after_wide && hipEventRecord(reinterpret_cast<hipEvent_t>(event_after_wide), st) != hipSuccess) return cbl_status() ? cbl_status() : CBL_ERR_BAD_ARG;
        return cbl_status();
    }
    // consumers of the WIDE result on other streams wait for this, not for the derivation and its tie replay
    if (event_after_wide && hipEventRecord(reinterpret_cast<hipEvent_t>(event_after_wide), st) != hipSuccess) return cbl_status() ? cbl_status() : CBL_ERR_BAD_ARG;
    {
        rc = launch_prefix(m, nsample_wide, nsample, idx_wide, dist2_wide, idx, dist2, tie_policy, worklist, counter, st);
        if (rc) return rc;
    }
    return cbl_knn_exact_worklist(b, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, worklist, counter, m, st, grids, cell_start, sorted);
}

CBL_EXPORT size_t cbl_knnquery_prefix_workspace_bytes(int m) { return m < 0 ? 0 : sizeof(int) * ((size_t)m + 64); }

CBL_EXPORT int cbl_knnquery_prefix(int b, int n, int m, int nsample_wide, int nsample, const float* xyz, const float* new_xyz,
                                   const int* offset, const int* new_offset, const int* idx_wide, const float* dist2_wide,
                                   int* idx, float* dist2, int tie_policy, void* workspace, size_t workspace_bytes, void* stream)
{
    (void)n;
    if (b <= 0 || m < 0 || nsample <= 0 || nsample >= nsample_wide || nsample_wide > CBL_KNN_MAX_NSAMPLE || tie_policy < 0 || tie_policy > 1) return CBL_ERR_BAD_ARG;
    if (m == 0) return CBL_OK;
    if (!xyz || !new_xyz || !offset || !new_offset || !idx_wide || !dist2_wide || !idx || !dist2 || !workspace) return CBL_ERR_BAD_ARG;
    if (workspace_bytes < cbl_knnquery_prefix_workspace_bytes(m)) return CBL_ERR_WORKSPACE;
    hipStream_t st = cbl_stream(stream);
    int* counter = reinterpret_cast<int*>(workspace);
    int* worklist = counter + 64;
    (void)hipMemsetAsync(counter, 0, sizeof(int), st);
    { const int rcp = launch_prefix(m, nsample_wide, nsample, idx_wide, dist2_wide, idx, dist2, tie_policy, worklist, counter, st); if (rcp) return rcp; }
    const int rc = cbl_status();
    if (rc) return rc;
    return cbl_knn_exact_worklist(b, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, worklist, counter, m, st, nullptr, nullptr, nullptr);
}
