// TEST INFRASTRUCTURE: runs the product's grid-search core (contrastboundary_amd/csrc/grid_core.h, the same
// __host__ __device__ code the HIP kernels instantiate) on the CPU, so its logic — cell assignment, shell
// traversal, the rounding-safe termination bound, tie certification — is checked against the oracle in the
// GPU-less build container.  The grid build here (bbox, counting sort) is a plain serial restatement of
// what knn_grid.hip does with atomics.
#include <algorithm>
#include <cstring>
#include <vector>
#include "../../contrastboundary_amd/csrc/grid_core.h"

static void build_grids(int b, const float* xyz, const int* offset, float pts_per_cell,
                        std::vector<CblGrid>& grids, std::vector<int>& cell_start, std::vector<float4>& sorted)
{
    grids.resize(b);
    int base = 0;
    for (int c = 0; c < b; c++) {
        const int s = c ? offset[c - 1] : 0, e = offset[c];
        float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        for (int i = s; i < e; i++)
            for (int a = 0; a < 3; a++) {
                const float v = xyz[3 * i + a];
                if (i == s || v < lo[a]) lo[a] = v;
                if (i == s || v > hi[a]) hi[a] = v;
            }
        CblGrid g; memset(&g, 0, sizeof g);
        const int cap = 2 * (e - s) + 64;
        cbl_grid_choose(g, lo, hi, e - s, pts_per_cell, cap);
        g.cell_base = base; g.start = s; g.end = e;
        base += cap;
        grids[c] = g;
    }
    const int n = b ? offset[b - 1] : 0;
    std::vector<int> cell(n), count(base + 1, 0);
    for (int c = 0; c < b; c++)
        for (int i = grids[c].start; i < grids[c].end; i++) {
            cell[i] = cbl_cell_of(grids[c], xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
            count[cell[i]]++;
        }
    cell_start.assign(base + 1, 0);
    for (int i = 0; i < base; i++) cell_start[i + 1] = cell_start[i] + count[i];
    std::vector<int> fill(cell_start.begin(), cell_start.end() - 1);
    sorted.resize(n);
    for (int i = n - 1; i >= 0; i--) {          // reverse: order inside a cell must not matter
        float4 v; v.x = xyz[3 * i]; v.y = xyz[3 * i + 1]; v.z = xyz[3 * i + 2]; v.w = cbl_as_float(i);
        sorted[fill[cell[i]]++] = v;
    }
}

template <int K>
static void run(int b, int m, int k, const float* new_xyz, const int* new_offset, const std::vector<CblGrid>& grids,
                const std::vector<int>& cell_start, const std::vector<float4>& sorted, int* idx, float* d2, int* certified)
{
    int c = 0;
    for (int q = 0; q < m; q++) {
        while (q >= new_offset[c]) c++;
        certified[q] = cbl_knn_grid_query<K>(grids[c], cell_start.data(), sorted.data(), new_xyz[3 * q], new_xyz[3 * q + 1],
                                             new_xyz[3 * q + 2], k, idx + (size_t)q * k, d2 + (size_t)q * k) ? 1 : 0;
        if (grids[c].end - grids[c].start <= K) certified[q] = 0;     // clouds with <= K supports always go to the exact kernel
    }
}

extern "C" int emul_knn_grid(int b, int m, int k, int ktemplate, float pts_per_cell, const float* xyz, const float* new_xyz,
                             const int* offset, const int* new_offset, int* idx, float* d2, int* certified)
{
    std::vector<CblGrid> grids; std::vector<int> cell_start; std::vector<float4> sorted;
    build_grids(b, xyz, offset, pts_per_cell, grids, cell_start, sorted);
    switch (ktemplate) {
        case 1: run<1>(b, m, k, new_xyz, new_offset, grids, cell_start, sorted, idx, d2, certified); break;
        case 4: run<4>(b, m, k, new_xyz, new_offset, grids, cell_start, sorted, idx, d2, certified); break;
        case 8: run<8>(b, m, k, new_xyz, new_offset, grids, cell_start, sorted, idx, d2, certified); break;
        case 16: run<16>(b, m, k, new_xyz, new_offset, grids, cell_start, sorted, idx, d2, certified); break;
        case 24: run<24>(b, m, k, new_xyz, new_offset, grids, cell_start, sorted, idx, d2, certified); break;
        case 36: run<36>(b, m, k, new_xyz, new_offset, grids, cell_start, sorted, idx, d2, certified); break;
        case 64: run<64>(b, m, k, new_xyz, new_offset, grids, cell_start, sorted, idx, d2, certified); break;
        default: return -1;
    }
    return 0;
}
