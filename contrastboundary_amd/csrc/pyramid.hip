// One layer of the ConvNet's input pyramid as ONE host call: the layer's own radius neighbourhoods, its grid-subsampled successor, the pooling
// and the upsampling indices.
//   tf_segmentation_inputs_radius   /root/reference/tensorflow/datasets/base.py:767-842 (loop body :795-812, last layer :815-820)
//     conv_i   = tf_batch_neighbors(points, points, lens, lens, r)            cropped to neighborhood_limits[i] (:756-765)
//     pool_p,b = tf_batch_subsampling(points, lens, sampleDl = 2 dl)
//     pool_i   = tf_batch_neighbors(pool_p, points, pool_b, lens, r)
//     up_i     = tf_batch_neighbors(points, pool_p, lens, pool_b, 2 r)
// The reference runs this inside tf.data workers (single-threaded C++ ops glued by Python).  Here the same sequence is issued by native host
// code over the kernels of tfops.hip / knn_grid.hip: ~30 launches and the one data-dependent host wait of a layer (the number of voxels the
// subsampling keeps) without an interpreter in between, so a loader thread can build the pyramid of the next scene (this call holds no Python
// lock) while the training thread issues the current scene's layers.  Values are those of the separate calls, bit for bit: the same
// kernels in the same order (tests/test_gpu_tfops.py compares with the op-by-op builder).
#include "cbl_common.h"
#include "../../include/cbl_amd.h"

namespace {

// offsets[c] = lengths[0] + ... + lengths[c]  (b is the number of clouds of a batch: a few)
__global__ void py_offsets_kernel(int b, const int* __restrict__ lengths, int* __restrict__ offsets)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int s = 0;
        for (int c = 0; c < b; c++) { s += lengths[c]; offsets[c] = s; }
    }
}

struct PyLayout { size_t off_q, off_p, total, counts, sub, end; };

PyLayout py_layout(int b, int n)
{
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    PyLayout L;
    size_t o = 0;
    L.off_q = o; o = up(o + sizeof(int) * (size_t)b);               // offsets of the layer's points
    L.off_p = o; o = up(o + sizeof(int) * (size_t)b);               // offsets of the sub-sampled points
    L.total = o; o = up(o + sizeof(int));                            // number of sub-sampled points
    L.counts = o; o = up(o + sizeof(int) * (size_t)(n > 0 ? n : 1));   // per-query counts of the search in flight (not returned)
    L.sub = o; o = up(o + cbl_grid_subsampling_workspace_bytes(b, n));
    L.end = o;
    return L;
}

}  // namespace

CBL_EXPORT size_t cbl_pyramid_layer_workspace_bytes(int b, int n)
{
    if (b <= 0 || n < 0) return 0;
    return py_layout(b, n).end;
}

// points (n,3), lengths (b) device; radius r of this layer, sample_dl = the next layer's cell (0: last layer — only `neighbors` is produced);
// grid_ws: workspace of cbl_radius_neighbors_workspace_bytes(b, n) bytes that holds (grid_is_built != 0) or will hold the search grid of
// (points, r); next_grid_ws: the same for (pool_points, 2 r), capacity n points, built by this call.
// Outputs: neighbors (n, limit); pool_points (capacity n, 3), pool_lengths (b), pools (capacity n, limit), upsamples (n, limit);
// max_counts (3, device): largest neighbourhood of the three searches (the reference's output widths before the crop);
// host_pool_points (host): the number of sub-sampled points, valid when the call returns (the call waits for it, and for nothing else).
CBL_EXPORT int cbl_pyramid_layer(int b, int n, const float* points, const int* lengths, float radius, float sample_dl, int limit,
                                 void* grid_ws, size_t grid_ws_bytes, int grid_is_built,
                                 int* neighbors, float* pool_points, int* pool_lengths, int* pools, int* upsamples,
                                 void* next_grid_ws, size_t next_grid_ws_bytes, int* max_counts, int* host_pool_points,
                                 void* workspace, size_t workspace_bytes, void* stream)
{
    if (b <= 0 || n < 0 || limit <= 0 || limit > 64 || !(radius > 0.f) || sample_dl < 0.f) return CBL_ERR_BAD_ARG;
    if (!points || !lengths || !grid_ws || !neighbors || !max_counts || !workspace) return CBL_ERR_BAD_ARG;
    const bool last = !(sample_dl > 0.f);
    if (!last && (!pool_points || !pool_lengths || !pools || !upsamples || !next_grid_ws || !host_pool_points)) return CBL_ERR_BAD_ARG;
    const PyLayout L = py_layout(b, n);
    if (workspace_bytes < L.end) return CBL_ERR_WORKSPACE;
    char* ws = reinterpret_cast<char*>(workspace);
    int* off_q = reinterpret_cast<int*>(ws + L.off_q);
    int* off_p = reinterpret_cast<int*>(ws + L.off_p);
    int* total = reinterpret_cast<int*>(ws + L.total);
    int* counts = reinterpret_cast<int*>(ws + L.counts);
    hipStream_t st = cbl_stream(stream);
    hipLaunchKernelGGL(py_offsets_kernel, dim3(1), dim3(64), 0, st, b, lengths, off_q);
    hipEvent_t sized = nullptr;
    int rc;
    if (!last) {
        // the subsampling first: its size travels to the host while the layer's own search runs
        rc = cbl_grid_subsampling(b, n, points, off_q, sample_dl, 0, nullptr, 0, nullptr, pool_points, nullptr, nullptr, pool_lengths, total,
                                  ws + L.sub, L.end - L.sub, stream);
        if (rc) return rc;
        if (hipMemcpyAsync(host_pool_points, total, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) return cbl_status();
        if (hipEventCreateWithFlags(&sized, hipEventDisableTiming) != hipSuccess) return cbl_status();
        if (hipEventRecord(sized, st) != hipSuccess) { hipEventDestroy(sized); return cbl_status(); }
    }
    rc = cbl_radius_neighbors_reuse(b, n, n, points, points, off_q, off_q, radius, limit, neighbors, counts, max_counts + 0, grid_ws, grid_ws_bytes,
                                    grid_is_built, stream);
    if (rc || last) { if (sized) hipEventDestroy(sized); return rc; }
    const hipError_t waited = hipEventSynchronize(sized);            // the one data-dependent host wait of the layer (the TF op's dynamic output shape)
    hipEventDestroy(sized);
    if (waited != hipSuccess) return cbl_status();
    const int m = *host_pool_points;
    if (m < 0 || m > n) return CBL_ERR_BAD_ARG;
    hipLaunchKernelGGL(py_offsets_kernel, dim3(1), dim3(64), 0, st, b, pool_lengths, off_p);
    // pooling: the sub-sampled points look for the layer's points within r (the grid of the self search, built above or earlier)
    rc = cbl_radius_neighbors_reuse(b, m, n, pool_points, points, off_p, off_q, radius, limit, pools, counts, max_counts + 1, grid_ws, grid_ws_bytes,
                                    (grid_is_built || n > 0) ? 1 : 0, stream);
    if (rc) return rc;
    // upsampling: the layer's points look for the sub-sampled points within 2 r; this builds the next layer's grid
    rc = cbl_radius_neighbors_reuse(b, n, m, points, pool_points, off_q, off_p, 2.0f * radius, limit, upsamples, counts, max_counts + 2, next_grid_ws,
                                    next_grid_ws_bytes, 0, stream);
    return rc;
}

// The whole pyramid in one call: cbl_pyramid_layer for layer 0 .. num_layers-1, every layer's outputs at the capacity of layer 0 (n rows; a layer never has
// more points than the one above it), the sizes handed from layer to layer on the host.  One call per scene means a loader thread needs the interpreter
// twice per pyramid (allocate, trim) instead of a dozen times.  Arrays are indexed by layer: neighbors[l] (n, limits[l]); pool_points[l] (n, 3),
// pool_lengths[l] (b), pools[l] (n, limits[l]), upsamples[l] (n, limits[l]) for l < num_layers - 1; grid_ws[l] one search-grid workspace per layer;
// max_counts (3 num_layers, device); host_sizes (num_layers, host — pinned recommended): points per layer, host_sizes[0] = n.
CBL_EXPORT int cbl_pyramid(int b, int n, const float* points, const int* lengths, float radius0, float dl0, int num_layers, const int* limits,
                           void* const* grid_ws, size_t grid_ws_bytes, int* const* neighbors, float* const* pool_points, int* const* pool_lengths,
                           int* const* pools, int* const* upsamples, int* max_counts, int* host_sizes, void* workspace, size_t workspace_bytes, void* stream)
{
    if (num_layers <= 0 || num_layers > 16 || !limits || !grid_ws || !neighbors || !max_counts || !host_sizes) return CBL_ERR_BAD_ARG;
    if (num_layers > 1 && (!pool_points || !pool_lengths || !pools || !upsamples)) return CBL_ERR_BAD_ARG;
    const float* pts = points;
    const int* lens = lengths;
    int n_l = n;
    float r = radius0, dl = dl0;
    host_sizes[0] = n;
    for (int l = 0; l < num_layers; l++) {
        const bool last = l == num_layers - 1;
        const int rc = cbl_pyramid_layer(b, n_l, pts, lens, r, last ? 0.f : 2.0f * dl, limits[l], grid_ws[l], grid_ws_bytes, l > 0 ? 1 : 0, neighbors[l],
                                         last ? nullptr : pool_points[l], last ? nullptr : pool_lengths[l], last ? nullptr : pools[l],
                                         last ? nullptr : upsamples[l], last ? nullptr : grid_ws[l + 1], last ? 0 : grid_ws_bytes, max_counts + 3 * l,
                                         last ? nullptr : host_sizes + l + 1, workspace, workspace_bytes, stream);
        if (rc) return rc;
        if (last) break;
        pts = pool_points[l]; lens = pool_lengths[l]; n_l = host_sizes[l + 1];
        r *= 2.0f; dl *= 2.0f;
    }
    return CBL_OK;
}

