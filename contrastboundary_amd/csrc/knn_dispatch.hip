// K1 front door: picks the grid kernels (K <= 64, certified, knn_grid.hip), the block-select kernel (64 < K <= 1024, certified,
// knn_select.hip) or the brute-force kernel (small problems, knn_exact.hip).
#include "cbl_common.h"

size_t cbl_knn_grid_workspace_bytes(int b, int n, int m, int nsample);     // knn_grid.hip
int cbl_knn_grid_launch(int b, int n, int m, int nsample, const float* xyz, const float* new_xyz, const int* offset,
                        const int* new_offset, int* idx, float* dist2, void* ws, size_t ws_bytes, int set_exact, hipStream_t st, int* order_out);

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
                         void* workspace, size_t workspace_bytes, int set_exact, void* stream, int* order_out = nullptr)
{
    if (b <= 0 || n < 0 || m < 0 || nsample <= 0 || nsample > CBL_KNN_MAX_NSAMPLE) return CBL_ERR_BAD_ARG;
    if (m == 0) return CBL_OK;
    if (!xyz || !new_xyz || !offset || !new_offset || !idx || !dist2) return CBL_ERR_BAD_ARG;
    const size_t need = cbl_knn_grid_workspace_bytes(b, n, m, nsample);
    if (need > 0 && workspace && workspace_bytes >= need)
        return cbl_knn_grid_launch(b, n, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, workspace, workspace_bytes, set_exact, cbl_stream(stream), order_out);
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
