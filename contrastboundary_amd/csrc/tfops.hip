// N1 / N3 / N2 / N4: the TF-side CPU ops of the reference on the GPU.
//   batch_grid_subsampling   /root/reference/tensorflow/ops/tf_custom_ops/tf_subsampling/grid_subsampling/grid_subsampling.cpp:6-161
//                            (TF op BatchGridSubsampling, tf_batch_subsampling.cpp:8-20)
//   grid_subsampling + features/labels   /root/reference/tensorflow/ops/cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-106
//   batch_nanoflann_neighbors            /root/reference/tensorflow/ops/tf_custom_ops/tf_neighbors/neighbors/neighbors.cpp:213-336  (kernel in knn_grid.hip)
//   cpp_knn_batch_omp                    /root/reference/tensorflow/ops/nearest_neighbors/knn_.cxx:104-135
//
// Grid subsampling, MI355X mapping: the reference walks the points once, single-threaded, through a hash map
// (unordered_map<size_t, SampledData>).  Here: one 64-bit key per point (cloud id in the top 16 bits, the reference's
// iX + NX*iY + NX*NY*iZ below), a STABLE radix sort of (key, point index) pairs (rocPRIM via hipCUB: the one library
// primitive in this path), head flags + scan to number the voxels, then one lane per voxel adds its points in INPUT order —
// the order the hash map accumulates them in — so barycentres are bit-identical to the reference; voxels come out in
// ascending key order per cloud (the reference's order is libstdc++'s hash iteration order, i.e. unspecified).
#include "cbl_common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

size_t cbl_radius_workspace_bytes_impl(int b, int ns);
int cbl_bbox_keys_launch(int b, int n, const float* xyz, const int* offset, unsigned* bbox, hipStream_t st);   // knn_grid.hip
int cbl_radius_launch(int b, int nq, int ns, const float* queries, const float* supports, const int* q_offset, const int* s_offset,
                      float radius, int limit, int* out, int* counts, int* max_count, void* ws, size_t ws_bytes, hipStream_t st, bool grid_is_built);

namespace {

__device__ __forceinline__ unsigned f2key(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float key2f(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

struct SubWs {
    unsigned* bbox;           // [6*b]
    unsigned long long* keys_in; unsigned long long* keys_out;     // [n]
    int* vals_in; int* vals_out;                                   // [n]
    int* flags; int* vox_id;                                       // [n]
    void* cub; size_t cub_bytes;
    size_t bytes;
};
inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

SubWs carve_sub(void* base, int b, int n)
{
    SubWs w;
    char* p = reinterpret_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) { char* r = p ? p + off : nullptr; off += up256(bytes); return r; };
    w.bbox = reinterpret_cast<unsigned*>(take(sizeof(unsigned) * 6 * (size_t)b));
    w.keys_in = reinterpret_cast<unsigned long long*>(take(8 * (size_t)n));
    w.keys_out = reinterpret_cast<unsigned long long*>(take(8 * (size_t)n));
    w.vals_in = reinterpret_cast<int*>(take(4 * (size_t)n));
    w.vals_out = reinterpret_cast<int*>(take(4 * (size_t)n));
    w.flags = reinterpret_cast<int*>(take(4 * (size_t)n));
    w.vox_id = reinterpret_cast<int*>(take(4 * (size_t)n));
    size_t sort_bytes = 0, scan_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, w.keys_in, w.keys_out, w.vals_in, w.vals_out, (size_t)(n > 0 ? n : 1));
    (void)rocprim::inclusive_scan(nullptr, scan_bytes, w.flags, w.vox_id, (size_t)(n > 0 ? n : 1), rocprim::plus<int>());
    w.cub_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
    w.cub = take(w.cub_bytes + 256);
    w.bytes = off;
    return w;
}

__global__ void sub_init_kernel(int b, unsigned* __restrict__ bbox, int* __restrict__ out_lengths, int* __restrict__ out_total)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 6 * b) bbox[i] = ((i % 6) < 3) ? 0xffffffffu : 0u;
    if (i < b) out_lengths[i] = 0;
    if (i == 0) out_total[0] = 0;
}

// key = cloud << 48 | (iX + NX*iY + NX*NY*iZ), all quantities computed with the reference's float expressions (:28-32, :61-64)
__global__ __launch_bounds__(256) void sub_keys_kernel(int b, int n, float dl, const float* __restrict__ pts, const int* __restrict__ offset,
                                                       const unsigned* __restrict__ bbox, unsigned long long* __restrict__ keys, int* __restrict__ vals)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int c = cbl_cloud_of(i, offset, b);
        const float inv = 1 / dl;
        float org[3], mx[3];
#pragma unroll
        for (int a = 0; a < 3; a++) { org[a] = floorf(key2f(bbox[6 * c + a]) * inv) * dl; mx[a] = key2f(bbox[6 * c + 3 + a]); }
        const unsigned long long NX = (unsigned long long)floorf((mx[0] - org[0]) / dl) + 1ull;
        const unsigned long long NY = (unsigned long long)floorf((mx[1] - org[1]) / dl) + 1ull;
        const unsigned long long iX = (unsigned long long)floorf((pts[3 * i + 0] - org[0]) / dl);
        const unsigned long long iY = (unsigned long long)floorf((pts[3 * i + 1] - org[1]) / dl);
        const unsigned long long iZ = (unsigned long long)floorf((pts[3 * i + 2] - org[2]) / dl);
        keys[i] = ((unsigned long long)c << 48) | ((iX + NX * iY + NX * NY * iZ) & 0xffffffffffffull);
        vals[i] = i;
    }
}

__global__ __launch_bounds__(256) void sub_flags_kernel(int n, const unsigned long long* __restrict__ keys, int* __restrict__ flags)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// one lane per voxel head: add the voxel's points in input order (stable sort => ascending original index)
__global__ __launch_bounds__(256) void sub_reduce_kernel(int n, const unsigned long long* __restrict__ keys, const int* __restrict__ vals,
                                                         const int* __restrict__ flags, const int* __restrict__ vox_id,
                                                         const float* __restrict__ pts, int fdim, const float* __restrict__ feat,
                                                         int ldim, const int* __restrict__ lab,
                                                         float* __restrict__ out_pts, float* __restrict__ out_feat, int* __restrict__ out_lab,
                                                         int* __restrict__ out_lengths, int* __restrict__ out_total)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        if (!flags[i]) continue;
        const unsigned long long key = keys[i];
        const int v = vox_id[i] - 1;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        int e = i;
        for (; e < n && keys[e] == key; e++) { const int p = vals[e]; sx += pts[3 * p]; sy += pts[3 * p + 1]; sz += pts[3 * p + 2]; }   // :76-84
        const int count = e - i;
        const float rc = (float)(1.0 / (double)count);                                       // point * (1.0 / count), :95
        out_pts[3 * v] = sx * rc; out_pts[3 * v + 1] = sy * rc; out_pts[3 * v + 2] = sz * rc;
        for (int c = 0; c < fdim; c++) {
            float fs = 0.f;
            for (int t = i; t < e; t++) fs += feat[(size_t)vals[t] * fdim + c];
            out_feat[(size_t)v * fdim + c] = fs / (float)count;                              // wrapper flavour :88-96
        }
        for (int c = 0; c < ldim; c++) {                                                      // majority vote; ties -> smallest label (canonical)
            int best = 0, bestcnt = -1;
            for (int t = i; t < e; t++) {
                const int l = lab[(size_t)vals[t] * ldim + c];
                int cnt = 0;
                for (int u = i; u < e; u++) cnt += lab[(size_t)vals[u] * ldim + c] == l;
                if (cnt > bestcnt || (cnt == bestcnt && l < best)) { best = l; bestcnt = cnt; }
            }
            out_lab[(size_t)v * ldim + c] = best;
        }
        // voxel counts without same-address atomics: the last head of a cloud publishes the running voxel number at the cloud's end
        const bool last_of_cloud = (e == n) || ((keys[e] >> 48) != (key >> 48));
        if (last_of_cloud) out_lengths[(int)(key >> 48)] = v + 1;          // cumulative for now; differenced by sub_lengths_kernel
        if (e == n) out_total[0] = v + 1;
    }
}

// cumulative voxel counts at cloud ends -> per-cloud counts (clouds without points keep 0)
__global__ void sub_lengths_kernel(int b, int* __restrict__ out_lengths)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int prev = 0;
    for (int c = 0; c < b; c++) { const int cum = out_lengths[c]; if (cum > 0) { out_lengths[c] = cum - prev; prev = cum; } }
}

// N4: local int64 indices of a dense batch from the stacked KNN result
__global__ __launch_bounds__(256) void knn_to_local_kernel(long long total, int per_batch_rows, int K, int npts, const int* __restrict__ idx, long long* __restrict__ out)
{
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long row = e / K;
        const int bb = (int)(row / per_batch_rows);
        out[e] = (long long)idx[e] - (long long)bb * npts;
    }
}


// ---- a13: dataloader voxelisation, /root/reference/pytorch/util/voxelize.py:4-16 (fnv_hash_vec), :38-56 (voxelize) --------
// key = FNV hash of floor(coord / voxel_size) taken as uint64 per axis (the reference multiplies by the prime THEN xors), argsort by
// key, run lengths per voxel.  The reference's np.argsort is quicksort (order inside a voxel unspecified); here the sort is stable,
// i.e. ascending original index inside a voxel.
template <typename T>
__global__ __launch_bounds__(256) void voxel_keys_kernel(int n, const T* __restrict__ coord, T voxel, unsigned long long* __restrict__ keys, int* __restrict__ vals)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        unsigned long long h = 14695981039346656037ull;                                  // :12
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const T d = floor(coord[3 * (size_t)i + a] / voxel);                         // np.floor(coord / voxel_size), :39
            h *= 1099511628211ull;                                                       // :14
            h ^= (unsigned long long)(long long)d;                                       // astype(np.uint64) of a non-negative float, :11,:15
        }
        keys[i] = h; vals[i] = i;
    }
}

// per voxel head: start position and count (count[v], start[v]); total voxels
__global__ __launch_bounds__(256) void voxel_runs_kernel(int n, const unsigned long long* __restrict__ keys, const int* __restrict__ flags,
                                                         const int* __restrict__ vox_id, int* __restrict__ start, int* __restrict__ count, int* __restrict__ total)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        if (!flags[i]) continue;
        const int v = vox_id[i] - 1;
        int e = i + 1;
        while (e < n && !flags[e]) e++;
        start[v] = i; count[v] = e - i;
        if (e == n) total[0] = v + 1;
    }
}

// order-preserving key of the squared distance to a centre point (data_util.py:62-64: argsort(sum(square(coord - coord_init), 1)))
template <typename T>
__global__ __launch_bounds__(256) void crop_keys_kernel(int n, const T* __restrict__ coord, int center, unsigned long long* __restrict__ keys, int* __restrict__ vals)
{
    const T cx = coord[3 * (size_t)center], cy = coord[3 * (size_t)center + 1], cz = coord[3 * (size_t)center + 2];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const T dx = coord[3 * (size_t)i] - cx, dy = coord[3 * (size_t)i + 1] - cy, dz = coord[3 * (size_t)i + 2] - cz;
        const double d = (double)((dx * dx + dy * dy) + dz * dz);                        // np.sum over 3 elements: sequential
        keys[i] = (unsigned long long)__double_as_longlong(d);                           // d >= 0: bit pattern is monotone
        vals[i] = i;
    }
}

}  // namespace

CBL_EXPORT size_t cbl_grid_subsampling_workspace_bytes(int b, int n) { return (b > 0 && n >= 0) ? carve_sub(nullptr, b, n).bytes : 0; }

CBL_EXPORT int cbl_grid_subsampling(int b, int n, const float* points, const int* offset, float dl,
                                    int fdim, const float* features, int ldim, const int* labels,
                                    float* out_points, float* out_features, int* out_labels, int* out_lengths, int* out_total,
                                    void* workspace, size_t workspace_bytes, void* stream)
{
    if (b <= 0 || b > 65535 || n < 0 || !(dl > 0.f) || fdim < 0 || ldim < 0) return CBL_ERR_BAD_ARG;
    if (!offset || !out_lengths || !out_total) return CBL_ERR_BAD_ARG;
    hipStream_t st = cbl_stream(stream);
    SubWs w = carve_sub(workspace, b, n);
    if (!workspace || workspace_bytes < w.bytes) return CBL_ERR_WORKSPACE;
    hipLaunchKernelGGL(sub_init_kernel, dim3(cbl_div_up(6 * b, 256)), dim3(256), 0, st, b, w.bbox, out_lengths, out_total);
    if (n == 0) return cbl_status();
    if (!points || !out_points || (fdim && (!features || !out_features)) || (ldim && (!labels || !out_labels))) return CBL_ERR_BAD_ARG;
    const dim3 g(cbl_grid_for(n, 256, 1024)), blk(256);
    int rc = cbl_bbox_keys_launch(b, n, points, offset, w.bbox, st);
    if (rc) return rc;
    hipLaunchKernelGGL(sub_keys_kernel, g, blk, 0, st, b, n, dl, points, offset, w.bbox, w.keys_in, w.vals_in);
    size_t cb = w.cub_bytes;
    hipError_t e = rocprim::radix_sort_pairs(w.cub, cb, w.keys_in, w.keys_out, w.vals_in, w.vals_out, (size_t)n, 0u, 64u, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(sub_flags_kernel, g, blk, 0, st, n, w.keys_out, w.flags);
    cb = w.cub_bytes;
    e = rocprim::inclusive_scan(w.cub, cb, w.flags, w.vox_id, (size_t)n, rocprim::plus<int>(), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(sub_reduce_kernel, g, blk, 0, st, n, w.keys_out, w.vals_out, w.flags, w.vox_id, points, fdim, features, ldim, labels,
                       out_points, out_features, out_labels, out_lengths, out_total);
    hipLaunchKernelGGL(sub_lengths_kernel, dim3(1), dim3(64), 0, st, b, out_lengths);
    return cbl_status();
}

CBL_EXPORT size_t cbl_radius_neighbors_workspace_bytes(int b, int ns) { return cbl_radius_workspace_bytes_impl(b, ns); }

CBL_EXPORT int cbl_radius_neighbors(int b, int nq, int ns, const float* queries, const float* supports, const int* q_offset, const int* s_offset,
                                    float radius, int limit, int* out, int* counts, int* max_count, void* workspace, size_t workspace_bytes, void* stream)
{
    if (b <= 0 || nq < 0 || ns < 0 || !(radius > 0.f) || limit <= 0 || limit > 64) return CBL_ERR_BAD_ARG;
    if (!max_count) return CBL_ERR_BAD_ARG;
    if (nq == 0) return (int)hipMemsetAsync(max_count, 0, sizeof(int), cbl_stream(stream));
    if (!queries || !supports || !q_offset || !s_offset || !out || !workspace) return CBL_ERR_BAD_ARG;
    return cbl_radius_launch(b, nq, ns, queries, supports, q_offset, s_offset, radius, limit, out, counts, max_count, workspace, workspace_bytes, cbl_stream(stream),
                             false);
}

// the same search over a workspace whose grid an earlier cbl_radius_neighbors / _reuse call built for the SAME supports, s_offset and radius (the pyramid
// builder searches every layer's points two or three times: as the supports of its own neighbourhoods, of the pooling and of the previous layer's
// upsampling, datasets/base.py:795-812): grid_is_built != 0 skips the 5-launch build
CBL_EXPORT int cbl_radius_neighbors_reuse(int b, int nq, int ns, const float* queries, const float* supports, const int* q_offset, const int* s_offset,
                                          float radius, int limit, int* out, int* counts, int* max_count, void* workspace, size_t workspace_bytes,
                                          int grid_is_built, void* stream)
{
    if (b <= 0 || nq < 0 || ns < 0 || !(radius > 0.f) || limit <= 0 || limit > 64) return CBL_ERR_BAD_ARG;
    if (!max_count) return CBL_ERR_BAD_ARG;
    if (nq == 0) return (int)hipMemsetAsync(max_count, 0, sizeof(int), cbl_stream(stream));      // (no grid is built: the next call must not claim one)
    if (!queries || !supports || !q_offset || !s_offset || !out || !workspace) return CBL_ERR_BAD_ARG;
    return cbl_radius_launch(b, nq, ns, queries, supports, q_offset, s_offset, radius, limit, out, counts, max_count, workspace, workspace_bytes, cbl_stream(stream),
                             grid_is_built != 0);
}

CBL_EXPORT int cbl_knn_indices_to_local(int B, int M, int K, int N, const int* idx, long long* out, void* stream)
{
    if (B < 0 || M < 0 || K <= 0 || N < 0) return CBL_ERR_BAD_ARG;
    const long long total = (long long)B * M * K;
    if (total == 0) return CBL_OK;
    if (!idx || !out) return CBL_ERR_BAD_ARG;
    hipLaunchKernelGGL(knn_to_local_kernel, dim3(cbl_grid_for(total, 256)), dim3(256), 0, cbl_stream(stream), total, M, K, N, idx, out);
    return cbl_status();
}

CBL_EXPORT size_t cbl_voxelize_workspace_bytes(int n) { return carve_sub(nullptr, 1, n > 0 ? n : 1).bytes; }

// coord (n,3) float32 (is_f64 = 0) or float64 (is_f64 = 1), non-negative (the caller subtracts the min, data_util.py:52-53)
// -> keys_sorted (n) u64, idx_sort (n) i32 [argsort, stable], start (n cap) / count (n cap) per voxel, num_voxels (1)
CBL_EXPORT int cbl_voxelize(int n, int is_f64, const void* coord, double voxel_size, unsigned long long* keys_sorted, int* idx_sort,
                            int* start, int* count, int* num_voxels, void* workspace, size_t workspace_bytes, void* stream)
{
    if (n < 0 || !(voxel_size > 0.0) || !num_voxels) return CBL_ERR_BAD_ARG;
    hipStream_t st = cbl_stream(stream);
    hipError_t e = hipMemsetAsync(num_voxels, 0, sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    if (n == 0) return CBL_OK;
    if (!coord || !keys_sorted || !idx_sort || !start || !count) return CBL_ERR_BAD_ARG;
    SubWs w = carve_sub(workspace, 1, n);
    if (!workspace || workspace_bytes < w.bytes) return CBL_ERR_WORKSPACE;
    const dim3 g(cbl_grid_for(n, 256, 1024)), blk(256);
    if (is_f64) hipLaunchKernelGGL(voxel_keys_kernel<double>, g, blk, 0, st, n, reinterpret_cast<const double*>(coord), voxel_size, w.keys_in, w.vals_in);
    else        hipLaunchKernelGGL(voxel_keys_kernel<float>, g, blk, 0, st, n, reinterpret_cast<const float*>(coord), (float)voxel_size, w.keys_in, w.vals_in);
    size_t cb = w.cub_bytes;
    e = rocprim::radix_sort_pairs(w.cub, cb, w.keys_in, keys_sorted, w.vals_in, idx_sort, (size_t)n, 0u, 64u, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(sub_flags_kernel, g, blk, 0, st, n, keys_sorted, w.flags);
    cb = w.cub_bytes;
    e = rocprim::inclusive_scan(w.cub, cb, w.flags, w.vox_id, (size_t)n, rocprim::plus<int>(), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(voxel_runs_kernel, g, blk, 0, st, n, keys_sorted, w.flags, w.vox_id, start, count, num_voxels);
    return cbl_status();
}

// data_util.py:62-64: indices of all points by ascending distance to coord[center] (stable); the caller keeps the first voxel_max
CBL_EXPORT int cbl_crop_order(int n, int is_f64, const void* coord, int center, int* order, void* workspace, size_t workspace_bytes, void* stream)
{
    if (n < 0 || center < 0 || (n > 0 && center >= n)) return CBL_ERR_BAD_ARG;
    if (n == 0) return CBL_OK;
    if (!coord || !order) return CBL_ERR_BAD_ARG;
    hipStream_t st = cbl_stream(stream);
    SubWs w = carve_sub(workspace, 1, n);
    if (!workspace || workspace_bytes < w.bytes) return CBL_ERR_WORKSPACE;
    const dim3 g(cbl_grid_for(n, 256, 1024)), blk(256);
    if (is_f64) hipLaunchKernelGGL(crop_keys_kernel<double>, g, blk, 0, st, n, reinterpret_cast<const double*>(coord), center, w.keys_in, w.vals_in);
    else        hipLaunchKernelGGL(crop_keys_kernel<float>, g, blk, 0, st, n, reinterpret_cast<const float*>(coord), center, w.keys_in, w.vals_in);
    size_t cb = w.cub_bytes;
    hipError_t e = rocprim::radix_sort_pairs(w.cub, cb, w.keys_in, w.keys_out, w.vals_in, order, (size_t)n, 0u, 64u, st);
    return e == hipSuccess ? cbl_status() : (int)e;
}
