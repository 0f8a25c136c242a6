// TEST INFRASTRUCTURE: C-ABI driver around the reference's OWN TF-side C++ cores, compiled from where they lie under
// /root/reference (oracle/Makefile target _ref/libref_tfops.so; nothing of the reference is copied into this repo):
//   batch_nanoflann_neighbors / batch_ordered_neighbors   tensorflow/ops/tf_custom_ops/tf_neighbors/neighbors/neighbors.cpp:125,213
//   batch_grid_subsampling / grid_subsampling              tensorflow/ops/tf_custom_ops/tf_subsampling/grid_subsampling/grid_subsampling.cpp:6,114
// (the functions the TF ops BatchOrderedNeighbors / BatchGridSubsampling call, tf_batch_neighbors.cpp:109, tf_batch_subsampling.cpp:96).
// Used to pin oracle/tfops_oracle.c and, through it, the HIP kernels.
#include <cstring>
#include <vector>
#include "/root/reference/tensorflow/ops/tf_custom_ops/tf_neighbors/neighbors/neighbors.h"
#include "/root/reference/tensorflow/ops/tf_custom_ops/tf_subsampling/grid_subsampling/grid_subsampling.h"

static std::vector<PointXYZ> to_pts(const float* p, int n)
{
    std::vector<PointXYZ> v((size_t)n);
    for (int i = 0; i < n; i++) v[i] = PointXYZ(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
    return v;
}

extern "C" {

// returns max_count; fills out (nq * max_count) if out != NULL and cap >= nq*max_count.  which: 0 nanoflann, 1 ordered (brute, sorted insert)
int ref_batch_neighbors(int which, int nq, const float* queries, int ns, const float* supports, int b, const int* q_batches, const int* s_batches,
                        float radius, int* out, long long cap)
{
    std::vector<PointXYZ> q = to_pts(queries, nq), s = to_pts(supports, ns);
    std::vector<int> qb(q_batches, q_batches + b), sb(s_batches, s_batches + b), res;
    if (which == 0) batch_nanoflann_neighbors(q, s, qb, sb, res, radius);
    else batch_ordered_neighbors(q, s, qb, sb, res, radius);
    const int max_count = nq ? (int)(res.size() / (size_t)nq) : 0;
    if (out && cap >= (long long)res.size()) memcpy(out, res.data(), res.size() * sizeof(int));
    return max_count;
}

// returns the number of subsampled points; out_points (cap*3), out_batches (b)
int ref_batch_grid_subsampling(int n, const float* points, int b, const int* batches, float dl, float* out_points, int* out_batches, int cap)
{
    std::vector<PointXYZ> p = to_pts(points, n), sub;
    std::vector<float> of, sf; std::vector<int> oc, sc;
    std::vector<int> ob(batches, batches + b), sb;
    batch_grid_subsampling(p, sub, of, sf, oc, sc, ob, sb, dl);
    if ((int)sub.size() <= cap)
        for (size_t i = 0; i < sub.size(); i++) { out_points[3 * i] = sub[i].x; out_points[3 * i + 1] = sub[i].y; out_points[3 * i + 2] = sub[i].z; }
    for (int i = 0; i < b; i++) out_batches[i] = sb[i];
    return (int)sub.size();
}

}  // extern "C"
