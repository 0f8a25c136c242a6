// The per-scene work of the ConvNet behind its input pyramid as ONE host call (BASELINE configs C5 / C3 per scene; round-4 review item 3): for every layer
//   AdaptiveWeight forward + backward     /root/reference/tensorflow/models/local_aggregation_operators.py:360-484 (under tf.gradients)
//   scene labels through the pools        /root/reference/tensorflow/models/heads/head.py:25-49 (get_scene_label_infer, reduction 'max')
//   contrast_head forward + backward      /root/reference/tensorflow/models/heads/head.py:462-807 ('softnn', 'l2', sample 'label')
// issued by native host code over the entry points a caller could use one by one (cbl_index_max, cbl_adaptive_weight_forward, cbl_neighbor_transpose,
// cbl_adaptive_weight_backward_csr, cbl_tf_scene_label, cbl_label_argmax, cbl_contrast_pairs_forward_samples, cbl_contrast_pairs_backward): the same
// kernels in the same order, ~70 launches without an interpreter, an autograd engine or an allocator in between (the Python-issued step spent as long
// issuing these launches as the device spends running them).  One transposed table per layer serves both backward passes (same neighbour table).
// Values are those of the separate calls bit for bit wherever those take the table as well (they use float atomics below 65536 pairs; this call
// always builds the table: deterministic).  Nothing here waits for the device.
#include "cbl_common.h"
#include "../../include/cbl_amd.h"

namespace {

__global__ __launch_bounds__(256) void cs_i64_to_i32_kernel(int n, const long long* __restrict__ in, int* __restrict__ out)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[i] = (int)in[i];
}
__global__ __launch_bounds__(256) void cs_i32_to_i64_kernel(int n, const int* __restrict__ in, long long* __restrict__ out)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[i] = (long long)in[i];
}
__global__ void cs_one_kernel(float* p) { p[0] = 1.f; }

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

// scratch of one layer inside the call's workspace
struct LayerWs { int* pad; int* inv_start; int* inv_src; void* tr_ws; size_t tr_bytes; void* aw_ws; size_t aw_bytes; float* per_point; float* stats; float* coef; float* own;
                 float* hist; long long* labels64; };

size_t carve_layer(char* base, size_t off, const CblConvnetLayer& L, int num_classes, bool need64, LayerWs* w)
{
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += up256(bytes); return p; };
    const size_t n = (size_t)L.n, nk = (size_t)L.n * (size_t)L.K;
    LayerWs t;
    t.pad = reinterpret_cast<int*>(take(sizeof(int)));
    t.inv_start = reinterpret_cast<int*>(take(sizeof(int) * (n + 1)));
    t.inv_src = reinterpret_cast<int*>(take(sizeof(int) * (nk > 0 ? nk : 1)));
    t.tr_bytes = cbl_neighbor_transpose_workspace_bytes(L.n, L.n, L.K);
    t.tr_ws = take(t.tr_bytes > 0 ? t.tr_bytes : 1);
    t.aw_bytes = cbl_adaptive_weight_backward_csr_workspace_bytes(L.n, L.n, L.C);
    t.aw_ws = take(t.aw_bytes > 0 ? t.aw_bytes : 1);
    t.per_point = reinterpret_cast<float*>(take(sizeof(float) * n));
    t.stats = reinterpret_cast<float*>(take(sizeof(float) * 2));
    t.coef = reinterpret_cast<float*>(take(sizeof(float) * (nk > 0 ? nk : 1)));
    t.own = reinterpret_cast<float*>(take(sizeof(float) * n * (size_t)(L.d > 0 ? L.d : 1)));
    t.hist = reinterpret_cast<float*>(take(sizeof(float) * n * (size_t)num_classes));
    t.labels64 = reinterpret_cast<long long*>(take(need64 ? sizeof(long long) * n : 1));
    if (w) *w = t;
    return off;
}

bool layer_ok(const CblConvnetLayer& L, int l)
{
    if (L.n < 1 || L.K < 1 || L.C < 4 || (L.C % 4) != 0 || L.d < 4) return false;
    if (!L.points || !L.neighbors || !L.features || !L.fc_weight || !L.fc_bias || !L.grad_out || !L.latent) return false;
    if (!L.aw_out || !L.grad_features || !L.grad_fc_weight || !L.grad_fc_bias || !L.cbl_loss || !L.cbl_mask || !L.grad_latent || !L.labels) return false;
    if (l > 0 && (!L.pools || L.Kp < 1)) return false;
    return true;
}

}  // namespace

CBL_EXPORT size_t cbl_convnet_step_workspace_bytes(int nlayers, const CblConvnetLayer* layers, int num_classes)
{
    if (nlayers < 1 || !layers || num_classes < 1) return 0;
    size_t off = up256(sizeof(float));                              // the upstream gradient of the losses (1.0)
    for (int l = 0; l < nlayers; l++) off = carve_layer(nullptr, off, layers[l], num_classes, l + 1 < nlayers, nullptr);
    return off;
}

CBL_EXPORT int cbl_convnet_step(int nlayers, const CblConvnetLayer* layers, const long long* point_labels, int num_classes, float temperature, float weight,
                                void* workspace, size_t workspace_bytes, void* stream)
{
    if (nlayers < 1 || !layers || !point_labels || num_classes < 1 || !workspace) return CBL_ERR_BAD_ARG;
    for (int l = 0; l < nlayers; l++)
        if (!layer_ok(layers[l], l)) return CBL_ERR_BAD_ARG;
    if (workspace_bytes < cbl_convnet_step_workspace_bytes(nlayers, layers, num_classes)) return CBL_ERR_WORKSPACE;
    hipStream_t st = cbl_stream(stream);
    char* base = static_cast<char*>(workspace);
    float* one = reinterpret_cast<float*>(base);
    size_t off = up256(sizeof(float));
    hipLaunchKernelGGL(cs_one_kernel, dim3(1), dim3(1), 0, st, one);
    const long long* labels_prev64 = point_labels;                  // labels of the layer above, int64 as cbl_tf_scene_label reads them
    int rc = 0;
    for (int l = 0; l < nlayers; l++) {
        const CblConvnetLayer& L = layers[l];
        LayerWs w;
        off = carve_layer(base, off, L, num_classes, l + 1 < nlayers, &w);
        // ---- AdaptiveWeight forward (reduction 'mean': the padding index is the largest entry of the table, local_aggregation_operators.py:466-470)
        if ((rc = cbl_index_max((long long)L.n * L.K, L.neighbors, w.pad, stream))) return rc;
        if ((rc = cbl_adaptive_weight_forward_ordered(L.n, L.n, L.K, L.C, L.points, L.points, L.neighbors, L.features, L.radius, L.fc_weight, L.fc_bias, w.pad, 1,
                                                      nullptr, L.aw_out, stream))) return rc;
        // ---- the layer's transposed neighbour table: AdaptiveWeight's backward and the contrast head's backward are gathers over it
        if ((rc = cbl_neighbor_transpose(L.n, L.n, L.K, L.neighbors, nullptr, nullptr, w.inv_start, w.inv_src, w.tr_ws, w.tr_bytes, stream))) return rc;
        if ((rc = cbl_adaptive_weight_backward_csr(L.n, L.n, L.K, L.C, L.points, L.points, L.neighbors, L.features, L.radius, L.fc_weight, L.fc_bias, w.pad, 1,
                                                   L.grad_out, nullptr, w.inv_start, w.inv_src, L.grad_features, L.grad_fc_weight, L.grad_fc_bias,
                                                   w.aw_ws, w.aw_bytes, stream))) return rc;
        // ---- scene labels of this layer: layer 0 holds the point labels, layer l the arg-max of the label histogram of its pooling neighbourhood
        if (l == 0) {
            hipLaunchKernelGGL(cs_i64_to_i32_kernel, dim3(cbl_grid_for(L.n, 256)), dim3(256), 0, st, L.n, point_labels, L.labels);
        } else {
            if ((rc = cbl_tf_scene_label(L.n, layers[l - 1].n, L.Kp, num_classes, labels_prev64, L.pools, 0, w.hist, stream))) return rc;
            if ((rc = cbl_label_argmax(L.n, num_classes, w.hist, L.labels, stream))) return rc;
        }
        if (l + 1 < nlayers) {
            hipLaunchKernelGGL(cs_i32_to_i64_kernel, dim3(cbl_grid_for(L.n, 256)), dim3(256), 0, st, L.n, L.labels, w.labels64);
            labels_prev64 = w.labels64;
        }
        // ---- contrast_head ('softnn', sample 'label') on the layer's radius neighbourhoods: mining + loss + coefficients, then the gradient
        if ((rc = cbl_contrast_pairs_forward_samples(L.n, L.n, 1, L.K, L.d, L.latent, L.labels, 0, 0.f, L.neighbors, nullptr, nullptr, nullptr, temperature, weight,
                                                     w.per_point, L.cbl_mask, w.stats, L.cbl_loss, w.coef, w.own, stream))) return rc;
        if ((rc = cbl_contrast_pairs_backward(L.n, L.K, L.d, L.latent, w.coef, w.own, nullptr, w.inv_start, w.inv_src, w.stats, one, weight, L.grad_latent, stream)))
            return rc;
    }
    return cbl_status();
}
