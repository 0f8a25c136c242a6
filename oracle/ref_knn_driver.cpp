// TEST INFRASTRUCTURE: C-ABI driver around the reference's kd-tree KNN, tensorflow/ops/nearest_neighbors/knn_.cxx:104-135
// (cpp_knn_batch_omp: what TF_OPS.tf_knn_search calls through the Cython module, tf_ops.py:111-129).
#include "/root/reference/tensorflow/ops/nearest_neighbors/knn_.h"

extern "C" void ref_knn_batch(const float* batch_data, long batch_size, long npts, const float* queries, long nqueries, long K, long* out, int omp)
{
    if (omp) cpp_knn_batch_omp(batch_data, (size_t)batch_size, (size_t)npts, 3, queries, (size_t)nqueries, (size_t)K, out);
    else cpp_knn_batch(batch_data, (size_t)batch_size, (size_t)npts, 3, queries, (size_t)nqueries, (size_t)K, out);
}

// single cloud: cpp_knn (knn_.cxx:22-44, one thread) / cpp_knn_omp (knn_.cxx:46-76, OpenMP over the queries) — the reference's own
// CPU KNN, timed by bench.py's cpu_baseline "reference" leg
extern "C" void ref_knn(const float* points, long npts, const float* queries, long nqueries, long K, long* out, int omp)
{
    if (omp) cpp_knn_omp(points, (size_t)npts, 3, queries, (size_t)nqueries, (size_t)K, out);
    else cpp_knn(points, (size_t)npts, 3, queries, (size_t)nqueries, (size_t)K, out);
}
