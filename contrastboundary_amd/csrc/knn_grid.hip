// K1 (fast): uniform-grid KNN with tie certification — placeholder until the grid kernel lands:
// reports "no workspace needed" so cbl_knnquery always takes the exact kernel.
#include "cbl_common.h"

size_t cbl_knn_grid_workspace_bytes(int, int, int, int) { return 0; }
int cbl_knn_grid_launch(int, int, int, int, const float*, const float*, const int*, const int*, int*, float*, void*, size_t, hipStream_t)
{
    return CBL_ERR_UNSUPPORTED;
}
