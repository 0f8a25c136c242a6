/*
 * cbl_amd.h — C ABI of libcbl_amd.so: the MI355X (gfx950) point-neighbourhood hot path of
 * LiyaoTang/contrastBoundary (pointops + CBL pair mining + TF-side neighbour/subsampling ops).
 *
 * Conventions (same as the reference boundary):
 *   - every pointer is a DEVICE pointer to contiguous row-major float32 / int32 data unless a
 *     parameter is documented as "host";
 *   - the caller owns and allocates every output and scratch buffer, and pre-zeroes the ones the
 *     reference pre-zeroes (documented per function); the callee borrows pointers for the duration
 *     of the asynchronous launch and never allocates or frees;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is enqueued on it
 *     and the call returns without synchronising;
 *   - stacked clouds: `offset[b]` int32 cumulative END offsets (pytorch side) or `lengths[b]` int32
 *     per-cloud lengths (TF side);
 *   - return value: 0 on success, CBL_ERR_* (<0) for rejected arguments, or a positive hipError_t.
 *
 * Each entry point cites the reference interface it replaces as path:line under /root/reference.
 * The reference launchers take no stream and return void; the extra leading `b` on the segmented
 * ops is the number of clouds (the reference recovers it by scanning `offset` linearly on device,
 * knnquery_cuda_kernel.cu:51-62).
 */
#ifndef CBL_AMD_H
#define CBL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CBL_OK               0
#define CBL_ERR_BAD_ARG     (-1)   /* null pointer / negative size / K out of range */
#define CBL_ERR_WORKSPACE   (-2)   /* caller-provided workspace too small */
#define CBL_ERR_UNSUPPORTED (-3)

#define CBL_KNN_MAX_NSAMPLE 1024   /* knnquery_cuda_kernel.cu:89-90: float best_dist[1024] */

/* library / build identification (host only) */
const char* cbl_version(void);            /* e.g. "cbl_amd 0.1 gfx950" */
int         cbl_device_arch_ok(void);     /* 1 if the current HIP device is gfx950, 0 otherwise, <0 no device */

/* ------------------------------------------------------------------------------------------------
 * pytorch/lib/pointops — the 10 exported entry points (pointops_api.cpp:12-23)
 * ---------------------------------------------------------------------------------------------- */

/* K1  knnquery_cuda_launcher  pytorch/lib/pointops/src/knnquery/knnquery_cuda_kernel.h:13
 *     kernel: knnquery_cuda_kernel.cu:65-111.
 * Segmented exact KNN: for query q in cloud c, scan supports [start_c, end_c) in index order with a
 * size-nsample max-heap (strict '<' replacement), then heap-sort ascending. Bit-identical idx/dist2
 * including tie order and the (1e10, start) sentinels when n_c < nsample.
 * Dispatches to the uniform-grid kernel (certified, with exact replay of tied queries) when
 * `workspace` is large enough (cbl_knnquery_workspace_bytes) and to the brute-force kernel otherwise
 * (workspace may be NULL / 0).
 *   xyz (n,3) new_xyz (m,3) offset (b) new_offset (b) -> idx (m,nsample) i32, dist2 (m,nsample) f32 */
int cbl_knnquery(int b, int n, int m, int nsample,
                 const float* xyz, const float* new_xyz,
                 const int* offset, const int* new_offset,
                 int* idx, float* dist2,
                 void* workspace, size_t workspace_bytes, void* stream);
size_t cbl_knnquery_workspace_bytes(int b, int n, int m, int nsample);

/* Same neighbour SET per query as cbl_knnquery, rows ascending by distance, but supports at exactly equal distance may appear in
 * any order (no replay for ties inside the list; ties at the K-th boundary are still resolved exactly like the reference, and so is a
 * tie for COLUMN 0: between two equidistant nearest supports — a query and a point coincident with it — the reference's choice stands).
 * For consumers that are invariant to the order of equal-distance neighbours: the CBL head (heads.py:192-199 drops column 0 — "the
 * query itself" — and reduces over the rest) and the sub-scene label mean (basic_operators.py:30-41). */
int cbl_knnquery_set(int b, int n, int m, int nsample,
                     const float* xyz, const float* new_xyz,
                     const int* offset, const int* new_offset,
                     int* idx, float* dist2,
                     void* workspace, size_t workspace_bytes, void* stream);
/* Third tie policy, for callers that are invariant to the order AND the choice among equally distant neighbours (every consumer in the
 * reference's networks is: softmax / max / mean over the K set): exact K smallest distances, ascending, but among supports tied at
 * the K-th distance those with the smallest indices are kept, and equal distances are listed by index: the K smallest by
 * (distance, index), a deterministic rule of its own.  Never replays a query through the reference's heap, so its cost
 * does not depend on ties — on coordinate grids / millimetre-quantised scans (many exactly equal distances) the reference-order
 * variants can spend milliseconds re-running tied queries one wave at a time.  dist2 is identical to cbl_knnquery's. */
int cbl_knnquery_anytie(int b, int n, int m, int nsample,
                        const float* xyz, const float* new_xyz,
                        const int* offset, const int* new_offset,
                        int* idx, float* dist2,
                        void* workspace, size_t workspace_bytes, void* stream);

/* cbl_knnquery / _set / _anytie (tie_policy 0 / 1 / 2) that also returns the CELL ORDER of the supports: cell_order[t] (n ints) = index
 * of the t-th support when the supports are listed cell by cell of the search grid (x fastest, then y, z; clouds one after the other) —
 * a by-product of the grid build (grid_scatter_kernel), written at no extra cost.  It is the processing order the *_ordered consumers
 * take.  Only the grid path produces it: CBL_ERR_UNSUPPORTED where cbl_knnquery would use another kernel (nsample > 64, n < 2048). */
int cbl_knnquery_ordered(int b, int n, int m, int nsample,
                         const float* xyz, const float* new_xyz,
                         const int* offset, const int* new_offset,
                         int* idx, float* dist2, int tie_policy, int* cell_order,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Measurement support (no counterpart in the reference): count[q] (n ints) = number of candidate supports in the 27-cell block around query q of the grid
 * search that LAST used `workspace` for a self-search over these (b, n, nsample) — the pairs that search evaluated in its first round ("pairs visited",
 * SURVEY 8(d)); the brute-force kernel it replaces (knnquery_cuda_kernel.cu:65-111) visits n_cloud pairs per query.  CBL_ERR_UNSUPPORTED off the grid path. */
int cbl_knn_grid_block_candidates(int b, int n, int nsample, const int* offset, int* count, void* workspace, size_t workspace_bytes, void* stream);

/* A narrower search derived from a wider one over the SAME supports and queries: idx_wide / dist2_wide (m, nsample_wide) from
 * cbl_knnquery (any tie policy), nsample < nsample_wide.  Rows whose first nsample entries are decided by the distances alone are copied;
 * rows with a tie that matters under tie_policy (0: reference order, 1: reference set — as cbl_knnquery / cbl_knnquery_set) or that are
 * not full are re-run through the reference-order replay, so idx / dist2 (m, nsample) equal what cbl_knnquery(nsample) returns, bit for
 * bit.  One search instead of two where a network looks at one geometry with several neighbourhood sizes (pointops.neighbor_cache
 * hints).  workspace: cbl_knnquery_prefix_workspace_bytes(m). */
size_t cbl_knnquery_prefix_workspace_bytes(int m);
int cbl_knnquery_prefix(int b, int n, int m, int nsample_wide, int nsample,
                        const float* xyz, const float* new_xyz, const int* offset, const int* new_offset,
                        const int* idx_wide, const float* dist2_wide, int* idx, float* dist2, int tie_policy,
                        void* workspace, size_t workspace_bytes, void* stream);

/* cbl_knnquery_ordered(nsample_wide, tie_policy_wide) followed by cbl_knnquery_prefix(nsample, tie_policy) in one call (cell_order may be
 * NULL): the derivation reuses the search's scratch, no separate workspace or counter reset; for nsample_wide > 16 the search kernel
 * itself writes the narrow rows and lists those a tie decides (no derivation pass).  event_after_wide (a hipEvent_t, or NULL) is
 * recorded on the stream behind the wide search and before the narrow rows' tie replay: what a consumer of the wide result on another
 * stream waits for.
 * CBL_ERR_UNSUPPORTED where the wide search
 * would not take the grid path — call the two functions instead. */
int cbl_knnquery_nested(int b, int n, int m, int nsample_wide, int tie_policy_wide, int nsample, int tie_policy,
                        const float* xyz, const float* new_xyz, const int* offset, const int* new_offset,
                        int* idx_wide, float* dist2_wide, int* idx, float* dist2, int* cell_order, void* event_after_wide,
                        void* workspace, size_t workspace_bytes, void* stream);

/* brute-force variant only (always bit-exact, O(m*n)); `algo` for tests/bench: see cbl_knnquery */
int cbl_knnquery_exact(int b, int n, int m, int nsample,
                       const float* xyz, const float* new_xyz,
                       const int* offset, const int* new_offset,
                       int* idx, float* dist2, void* stream);

/* K2  furthestsampling_cuda_launcher  sampling/sampling_cuda_kernel.h:12 ; kernel .cu:14-129.
 * `n_max` = longest cloud (selects the reference block size B = 2^floor(log2 n_max) <= 1024 whose
 * tree reduction defines the tie rule).  tmp (n) must be pre-filled with 1e10 (pointops.py:22).
 *   xyz (n,3) offset (b) new_offset (b) tmp (n) -> idx (new_offset[b-1]) i32 */
int cbl_furthestsampling(int b, int n_max, const float* xyz, const int* offset, const int* new_offset,
                         float* tmp, int* idx, void* stream);
/* Same op, same samples, with scratch for the spatial-bucket kernel (fps_bucket.hip) that large clouds take: `n` = total rows
 * (= offset[b-1]); cbl_furthestsampling_workspace_bytes() == 0 means the plain entry is used and workspace may be NULL. */
size_t cbl_furthestsampling_workspace_bytes(int b, int n, int n_max);
int cbl_furthestsampling_ws(int b, int n, int n_max, const float* xyz, const int* offset, const int* new_offset,
                            float* tmp, int* idx, void* workspace, size_t workspace_bytes, void* stream);
/* The sampler for CHAINS of samplings (the network samples 40960 -> 10240 -> 2560 -> 640 -> 160, every stage from the previous stage's samples in
 * sampling order, pytorch/model/blocks.py:61-68): furthest point sampling of an FPS sequence is its prefix whenever every arg-max of the first run was
 * attained by one point only (with equal maxima the reference's rank rule decides, and ranks are positions).
 *   cert_out (b) i32: number of leading samples of every cloud that were unique maxima, counted up to half of the cloud's samples (a later stage asks
 *            for a fraction of them; 0 = no certificate: kernels without the check);
 *   cert_in  (b) i32 or NULL: the certificate of the run that PRODUCED xyz (xyz = that run's samples, in order, unmodified).  A cloud with
 *            cert_in[c] >= its requested sample count gets idx = its first rows and cert_out[c] = cert_in[c] (tmp is left untouched for it);
 *            every other cloud is sampled as by cbl_furthestsampling_ws.
 * Both runs start from the reference wrapper's distances (tmp = 1e10 everywhere, pointops.py:16-24).  Exactness under ties: tests/test_gpu_pointops.py. */
int cbl_furthestsampling_chain(int b, int n, int n_max, const float* xyz, const int* offset, const int* new_offset, float* tmp, int* idx,
                               const int* cert_in, int* cert_out, void* workspace, size_t workspace_bytes, void* stream);

/* K3/K4  grouping_{forward,backward}_cuda_launcher  grouping/grouping_cuda_kernel.h:13-14.
 *   forward : input (n,c), idx (m,nsample) -> output (m,nsample,c)        (output fully overwritten)
 *   backward: grad_output (m,nsample,c), idx -> grad_input (n,c) +=       (caller pre-zeroes) */
int cbl_grouping_forward(int m, int nsample, int c, const float* input, const int* idx, float* output, void* stream);
/* same values; `order` (m ints, NULL = none) = processing sequence of the m query points, see cbl_queryandgroup_ordered */
int cbl_grouping_forward_ordered(int m, int nsample, int c, const float* input, const int* idx, const int* order, float* output, void* stream);
int cbl_grouping_backward(int m, int nsample, int c, const float* grad_output, const int* idx, float* grad_input, void* stream);

/* Transposed neighbour table ("CSR by target", SURVEY.md 7 hard part 6): for every target row the pairs (source, column) of
 * idx (m, nsample) that point at it.  Segment r = [inv_start[r], inv_start[r+1]) of inv_src lists, ASCENDING, the flat pair indices
 * p = source * nsample + column with idx[p] == order_dst[r] (== r without an order); entries outside [0, n) (shadow padding) are left out.
 * order_src (m) / order_dst (n): optional processing sequences of the sources / targets (the cell order of the search: the same array
 * for a self-search); they only make the build local, the table's CONTENT is defined by order_dst alone.  It turns every scatter-add
 * backward of the path (grouping_cuda_kernel.cu:16-25 and the index_select backward inside heads.py:185-246) into a gather with a segmented
 * sum in the reference loop's order: no atomics, no zero fill, deterministic.  n <= 1 M (CBL_ERR_UNSUPPORTED beyond); no global atomic per pair.
 *   -> inv_start (n+1) i32, inv_src (m*nsample) i32 (the first inv_start[n] entries are used) */
size_t cbl_neighbor_transpose_workspace_bytes(int m, int n, int nsample);
int cbl_neighbor_transpose(int m, int n, int nsample, const int* idx, const int* order_src, const int* order_dst, int* inv_start, int* inv_src,
                           void* workspace, size_t workspace_bytes, void* stream);
/* Two neighbour tables of ONE geometry (same m sources, n targets and orders — e.g. the K = 8 / 16 table of a stage's blocks and the K = 36 table of its CBL
 * head, blocks.py:34-35 / heads.py:190-196) transposed together: the outputs are those of two cbl_neighbor_transpose calls, by four launches instead of eight
 * (the build is a chain of small latency-bound kernels; the two tables' chains run side by side).  Tables beyond 131072 targets are built one after the other. */
size_t cbl_neighbor_transpose_pair_workspace_bytes(int m, int n, int nsample_a, int nsample_b);
int cbl_neighbor_transpose_pair(int m, int n, int nsample_a, const int* idx_a, int nsample_b, const int* idx_b, const int* order_src, const int* order_dst,
                                int* inv_start_a, int* inv_src_a, int* inv_start_b, int* inv_src_b, void* workspace, size_t workspace_bytes, void* stream);
/* K4 grouping_backward_cuda_launcher (grouping_cuda_kernel.h:14) as a gather over the transposed table of its idx:
 *   grad_input[order_dst[r], :] = sum over segment r of grad_output[p, :]   (written, not accumulated: no pre-zeroing; same summation order
 *   as the reference loop run sequentially, so bit-identical to the CPU oracle) */
int cbl_grouping_backward_csr(int n, int c, const float* grad_output, const int* order_dst, const int* inv_start, const int* inv_src,
                              float* grad_input, void* stream);
/* the same for rows that are a column slice of wider rows: pair p's row = grad_output[p * row_stride + col_offset .. + c) — the feature part
 * of queryandgroup's (m, nsample, 3 + c) gradient (pointops.py:90-98: torch.cat) without first copying the slice out */
int cbl_grouping_backward_csr_rows(int n, int c, int row_stride, int col_offset, const float* grad_output, const int* order_dst,
                                   const int* inv_start, const int* inv_src, float* grad_input, void* stream);
/* The remaining scatter-adds of the path as gathers over the same table (no atomics, ascending pair order = the reference loops run sequentially:
 * bit-exact against the CPU oracle, run-to-run deterministic).
 * cbl_weighted_scatter_csr: grad_input[t, ch] = sum over the pairs p = (source, column) listing t of rows[source, ch] * weight[p, ch % w_c]
 *   - K6, interpolation_cuda_kernel.cu:20-33 (rows = grad_output (m, c), weight (m, 3), nsample = 3, w_c = 1);
 *   - the grad_input part of K10, aggregation_cuda_kernel.cu:22-39 (rows = grad_output (n, c), weight (n, nsample, w_c)); the other two outputs
 *     come from cbl_aggregation_backward called with grad_input = NULL.
 * cbl_subtraction_backward_csr: K8, subtraction_cuda_kernel.cu:18-30: grad_input1 (m, c) accumulated (+= sum over s), grad_input2 (n2, c) written. */
int cbl_weighted_scatter_csr(int n, int nsample, int c, int w_c, const float* rows, const float* weight, const int* order_dst, const int* inv_start,
                             const int* inv_src, float* grad_input, void* stream);
int cbl_subtraction_backward_csr(int m, int n2, int nsample, int c, const float* grad_output, const int* order_dst, const int* inv_start,
                                 const int* inv_src, float* grad_input1, float* grad_input2, void* stream);

/* K5/K6  interpolation_{forward,backward}_cuda_launcher  interpolation/interpolation_cuda_kernel.h:13-14.
 *   forward : input (m,c), idx (n,k), weight (n,k) -> output (n,c) +=     (caller pre-zeroes)
 *   backward: grad_output (n,c), idx, weight -> grad_input (m,c) +=       (caller pre-zeroes) */
int cbl_interpolation_forward(int n, int c, int k, const float* input, const int* idx, const float* weight, float* output, void* stream);
int cbl_interpolation_backward(int n, int c, int k, const float* grad_output, const int* idx, const float* weight, float* grad_input, void* stream);

/* K7/K8  subtraction_{forward,backward}_cuda_launcher  subtraction/subtraction_cuda_kernel.h:13-14.
 *   forward : input1 (n,c), input2 (n,c), idx (n,nsample) -> output (n,nsample,c) = in1[n]-in2[idx]
 *   backward: grad_output (n,nsample,c) -> grad_input1 (n,c) +=, grad_input2 (n,c) +=  (pre-zeroed) */
int cbl_subtraction_forward(int n, int nsample, int c, const float* input1, const float* input2, const int* idx, float* output, void* stream);
/* same values; `order` (n ints, NULL = none) = processing sequence of the points, see cbl_queryandgroup_ordered */
int cbl_subtraction_forward_ordered(int n, int nsample, int c, const float* input1, const float* input2, const int* idx, const int* order,
                                    float* output, void* stream);
int cbl_subtraction_backward(int n, int nsample, int c, const int* idx, const float* grad_output, float* grad_input1, float* grad_input2, void* stream);

/* K9/K10 aggregation_{forward,backward}_cuda_launcher  aggregation/aggregation_cuda_kernel.h:13-14.
 *   forward : input (n,c), position (n,nsample,c), weight (n,nsample,w_c), idx (n,nsample)
 *             -> output (n,c) += sum_k (input[idx]+position)*weight[..., c % w_c]   (pre-zeroed)
 *   backward: -> grad_input (n,c) += (pre-zeroed), grad_position (n,nsample,c) = (overwritten),
 *             grad_weight (n,nsample,w_c) += (pre-zeroed) */
int cbl_aggregation_forward(int n, int nsample, int c, int w_c, const float* input, const float* position, const float* weight, const int* idx, float* output, void* stream);
/* same values; `order` (n ints, NULL = none) = processing sequence of the points, see cbl_queryandgroup_ordered */
int cbl_aggregation_forward_ordered(int n, int nsample, int c, int w_c, const float* input, const float* position, const float* weight, const int* idx,
                                    const int* order, float* output, void* stream);
int cbl_aggregation_backward(int n, int nsample, int c, int w_c, const float* input, const float* position, const float* weight, const int* idx, const float* grad_output, float* grad_input, float* grad_position, float* grad_weight, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused composites of the reference's python-level ops (same values, one launch)
 * ---------------------------------------------------------------------------------------------- */

/* F1  queryandgroup  pytorch/lib/pointops/functions/pointops.py:79-100 (idx given).
 *   xyz (n,3) new_xyz (m,3) feat (n,c) idx (m,nsample) -> out (m,nsample,3+c) if use_xyz else (m,nsample,c)
 *   out[m,k,0:3] = xyz[idx]-new_xyz[m]; out[m,k,3:] = feat[idx] */
int cbl_queryandgroup(int m, int nsample, int c, int use_xyz, const float* xyz, const float* new_xyz, const float* feat, const int* idx, float* out, void* stream);
/* Same result; `order` (m ints, a permutation of the query points; NULL = cbl_queryandgroup) is the SEQUENCE in which the points are
 * processed.  Given a spatially coherent sequence — the cell order of the neighbour search, cbl_knnquery_ordered — consecutive
 * workgroups gather neighbouring supports and each XCD is handed one contiguous eighth of the sequence, so the feature rows it reads
 * stay in its own L2 instead of being re-fetched through the fabric (scenes arrive shuffled: pointops.py has no such notion, the
 * argument changes the schedule, never the values).  Shapes the ordered kernel does not cover fall back to the plain one. */
int cbl_queryandgroup_ordered(int m, int nsample, int c, int use_xyz, const float* xyz, const float* new_xyz, const float* feat, const int* idx,
                              const int* order, float* out, void* stream);

/* F4  interpolation weights  pointops.py:171-174: w = (1/(dist+1e-8)) / sum_k(1/(dist+1e-8)), dist = sqrt(dist2)
 *   dist2 (n,k) -> weight (n,k), dist (n,k) (dist may be NULL) */
int cbl_interpolation_weights(int n, int k, const float* dist2, float* weight, float* dist, void* stream);


/* ------------------------------------------------------------------------------------------------
 * Contrastive Boundary Learning head (pytorch side)
 * ---------------------------------------------------------------------------------------------- */

/* a7  get_subscene_label / get_subscene_features  pytorch/model/basic_operators.py:9-50 (idx given)
 *   target (N) int64 class ids, neighbor_idx (m,kr) rows into target (from cbl_knnquery with nsample = kr,
 *   supports = stage-0 points) -> out (m,num_classes) = mean one-hot label of the kr neighbours.  num_classes <= 64 */
int cbl_subscene_label(int m, int kr, int num_classes, const long long* target, const int* neighbor_idx, float* out, void* stream);

/* torch.argmax(labels, -1) (first maximal index), as used by posmask_cnt  pytorch/model/heads.py:145-149
 *   labels (m,num_classes) -> amax (m) int32 */
int cbl_label_argmax(int m, int num_classes, const float* labels, int* amax, void* stream);

/* F5  ContrastHead.point_contrast  pytorch/model/heads.py:185-246 with pos='cnt', dist='l2', contrast='softnn'
 *   features (m,d) f32 (d in {4,8,16,32,64}, 16-byte aligned), amax (m) i32 = argmax of the (sub-scene) label,
 *   neighbor_idx (m,nsample) i32 from cbl_knnquery on the stage's own points (column 0 = self, dropped; nsample <= 65)
 *   -> per_point (m) f32 loss of each point (0 where masked), point_mask (m) i32 (1 = has both positive and negative
 *      neighbours), stats (2) f32 = {sum of per-point losses, #masked-in points}, loss (1) f32 = weight * mean
 *      (0 when no point qualifies, heads.py:233).  No host synchronisation. */
int cbl_point_contrast_forward(int m, int nsample, int d, const float* features, const int* amax, const int* neighbor_idx,
                               float temperature, float weight, float* per_point, int* point_mask, float* stats, float* loss, void* stream);
/* backward of the above w.r.t. features: grad_features (m,d) += grad_loss[0] * d loss / d features  (caller pre-zeroes) */
int cbl_point_contrast_backward(int m, int nsample, int d, const float* features, const int* amax, const int* neighbor_idx,
                                float temperature, float weight, const float* stats, const float* grad_loss, float* grad_features, void* stream);

/* Training path of F5 / a16: forward AND the gradient w.r.t. features in one pass over the neighbour rows.  grad_unit (m or n_valid
 * rows x d, caller pre-zeroes) receives the gradient without its global factor grad_loss * weight / #qualifying points, which only
 * exists after the launch (stats[1]); cbl_contrast_grad_scale applies it when the backward pass arrives:
 *   grad_features[e] = grad_unit[e] * grad_loss[0] * weight / stats[1]   (all zeros when stats[1] == 0). */
int cbl_point_contrast_forward_grad(int m, int nsample, int d, const float* features, const int* amax, const int* neighbor_idx,
                                    float temperature, float weight, float* per_point, int* point_mask, float* stats, float* loss,
                                    float* grad_unit, void* stream);
/* the two point-contrast entry points taking the hard labels as the reference holds them (torch.long, heads.py:186-189: the target
 * itself at stage 0) instead of an int32 copy: one conversion pass less per step.  Values in the int32 range (class ids, ignore labels). */
int cbl_point_contrast_forward_l64(int m, int nsample, int d, const float* features, const long long* labels, const int* neighbor_idx,
                                   float temperature, float weight, float* per_point, int* point_mask, float* stats, float* loss, void* stream);
int cbl_point_contrast_forward_grad_l64(int m, int nsample, int d, const float* features, const long long* labels, const int* neighbor_idx,
                                        float temperature, float weight, float* per_point, int* point_mask, float* stats, float* loss,
                                        float* grad_unit, void* stream);
int cbl_tf_contrast_forward_grad(int m, int n_valid, int nsample, int d, const float* features, const int* labels, const int* neighbors,
                                 float temperature, float weight, float* per_point, int* point_mask, float* stats, float* loss,
                                 float* grad_unit, void* stream);
/* TF contrast_head with sample 'labelkl<thr>' (tensorflow/models/heads/head.py:492-519, configs s3dis.py:162-163 — the README's
 * "ConvNet + CBL (kl)" row): soft_labels (n_valid, num_classes) f32 = the sub-scene label DISTRIBUTIONS (cbl_tf_scene_label with
 * by_valid = 1; one-hot at stage 0); neighbour j of point i is a positive iff sum_c xlogy(p_i[c], p_i[c] / max(p_j[c], 1e-12)) <
 * kl_threshold (a shadow neighbour gathers the zero row and only the shadow mask restricts the pairs).  Everything else as
 * cbl_tf_contrast_forward / _forward_grad.  num_classes <= 255. */
int cbl_tf_contrast_forward_kl(int m, int n_valid, int nsample, int d, const float* features, const float* soft_labels, int num_classes,
                               float kl_threshold, const int* neighbors, float temperature, float weight, float* per_point, int* point_mask,
                               float* stats, float* loss, void* stream);
int cbl_tf_contrast_forward_grad_kl(int m, int n_valid, int nsample, int d, const float* features, const float* soft_labels, int num_classes,
                                    float kl_threshold, const int* neighbors, float temperature, float weight, float* per_point, int* point_mask,
                                    float* stats, float* loss, float* grad_unit, void* stream);
int cbl_contrast_grad_scale(long long total, const float* grad_unit, const float* stats, const float* grad_loss, float weight,
                            float* grad_features, void* stream);

/* The same head with an atomic-free gradient (flavours by `flags`: bit 0 = TF contrast_head head.py:462-807, bit 1 = labels are int64;
 * num_classes > 0: `labels` are (n_valid, num_classes) f32 distributions and positives are KL(p_i || p_j) < kl_threshold, head.py:492-519).
 *   forward : per_point / point_mask / stats / loss as cbl_point_contrast_forward; with coef != NULL also the scalar coefficient of every
 *             pair, coef (m, nsample) (column 0 = 0), and the centre half of the gradient grad_own (m, d), both WITHOUT the global factor
 *             grad_loss * weight / count;  `order` (m, NULL = none) = processing sequence, values do not depend on it
 *   backward: grad_features[t] = (grad_own[t] + sum over the pairs p = (i, col) listing t of coef[p] (f_t - f_i)) * grad_loss * weight / count,
 *             a gather over the transposed table of neighbor_idx (cbl_neighbor_transpose with n = m, order_dst = order)
 * d in {4, 8, 16, 32, 64}, nsample <= 65. */
int cbl_contrast_pairs_forward(int m, int n_valid, int flags, int nsample, int d, const float* features, const void* labels, int num_classes,
                               float kl_threshold, const int* neighbor_idx, const int* order, float temperature, float weight,
                               float* per_point, int* point_mask, float* stats, float* loss, float* coef, float* grad_own, void* stream);
/* TF contrast_head with the sample strings beyond 'label' and the 'S' margin (tensorflow/models/heads/head.py:560-625 sample_labels, :759-760 and
 * :783-785 margin 'S'): as cbl_contrast_pairs_forward with flags bit 0 set, and
 *   sample_idx (m, nsample) = the self column followed by the '-'-concatenated sample columns (neighbour columns for 'label', the first k
 *       neighbours for 'nn<k>', the caller's random draws for 'rand<n>' — the reference draws them with tf.random.uniform per cloud, :579-596);
 *   roles (nsample - 1) u8 per column after the self column: 0 = mined from the labels (:598-600, valid mask :540-545), CBL_ROLE_POS = 'nn' (always a
 *       positive, also when it is a shadow neighbour: that gathers the zero row), CBL_ROLE_NEG = 'rand' (always a negative), CBL_ROLE_NEG_REJECT =
 *       'rand<n>R' (a negative unless sample_valid (m, nsample - 1) u8 is 0 there: the draw is one of the point's neighbours, :611-615); NULL = all 0;
 *   flags bit 3 = margin 'S': softnn pos / max(neg, eps) instead of pos / (pos + neg); nce: under_j = e_j + sum of the negatives instead of the sum of
 *       all valid exps.  (bit 2 = 'nce', bit 1 = int64 labels as before.)
 * coef / grad_own feed cbl_contrast_pairs_backward over the transposed table of sample_idx (or _backward_atomic) unchanged. */
#define CBL_ROLE_LABEL 0
#define CBL_ROLE_POS 1
#define CBL_ROLE_NEG 2
#define CBL_ROLE_NEG_REJECT 3
int cbl_contrast_pairs_forward_samples(int m, int n_valid, int flags, int nsample, int d, const float* features, const void* labels, int num_classes,
                                       float kl_threshold, const int* sample_idx, const unsigned char* roles, const unsigned char* sample_valid,
                                       const int* order, float temperature, float weight, float* per_point, int* point_mask, float* stats,
                                       float* loss, float* coef, float* grad_own, void* stream);
int cbl_contrast_pairs_backward(int m, int nsample, int d, const float* features, const float* coef, const float* grad_own, const int* order,
                                const int* inv_start, const int* inv_src, const float* stats, const float* grad_loss, float weight,
                                float* grad_features, void* stream);

/* the same gradient where no transposed table exists (more than 1 M rows): grad_own scaled, then one float atomic per (pair with a coefficient,
 * channel) — the scatter of the reference's index_select backward (heads.py:185-246 under autograd); last bits depend on the order of the atomics */
int cbl_contrast_pairs_backward_atomic(int m, int n_valid, int nsample, int d, const float* features, const float* coef, const float* grad_own,
                                       const int* neighbor_idx, const float* stats, const float* grad_loss, float weight,
                                       float* grad_features, void* stream);

/* a16  TF contrast_head  tensorflow/models/heads/head.py:462-807 with sample 'label', contrast 'softnn', dist 'l2' on RADIUS
 *      neighbourhoods (ids >= n_valid are the search's shadow padding; negative hard labels = ignored points):
 *   features (m,d), labels (n_valid >= m rows, i32 hard label per point, from point_labels or cbl_tf_scene_label + cbl_label_argmax),
 *   neighbors (m,nsample) i32 incl. the self column 0 (dropped, :560) -> per_point / point_mask / stats / loss as cbl_point_contrast_forward.
 *   Differences from the pytorch head, all reproduced: valid mask (:540-545), dist = sqrt(max(.,1e-12)) (:184-185), max-shift over all columns (:752). */
int cbl_tf_contrast_forward(int m, int n_valid, int nsample, int d, const float* features, const int* labels, const int* neighbors,
                            float temperature, float weight, float* per_point, int* point_mask, float* stats, float* loss, void* stream);
int cbl_tf_contrast_backward(int m, int n_valid, int nsample, int d, const float* features, const int* labels, const int* neighbors,
                             float temperature, float weight, const float* stats, const float* grad_loss, float* grad_features, void* stream);
/* get_scene_label_infer + get_neighbor_summary  head.py:25-49, :117-131: class histogram of point_labels over scene_neighbor (m,k) (pad = n_valid,
 * label < 0 ignored) -> out (m,num_classes): counts / k (by_valid = 0; argmax of it = reduction 'max') or counts / (#valid + 1e-12) ('soft') */
int cbl_tf_scene_label(int m, int n_valid, int k, int num_classes, const long long* point_labels, const int* scene_neighbor, int by_valid, float* out, void* stream);

/* a9  get_boundary_mask  pytorch/model/basic_operators.py:69-97 (labels (n) int64, negative = invalid neighbour label)
 *   neighbor_idx (n,k) -> bound (n) u8 [any valid neighbour label differs], plain (n) u8 [all valid neighbour labels equal],
 *   cnt (n) i32 [number of differing valid neighbours]; any output may be NULL */
int cbl_boundary_mask(int n, int k, const long long* labels, const int* neighbor_idx, unsigned char* bound, unsigned char* plain, int* cnt, void* stream);

/* cumulate_probs  pytorch/tool/test.py:330-352: the test loop's accumulation of a batch of crops' predictions into the cloud's per-point rows.
 *   probs (n, num_classes) f32, inds (m) i64 point of every prediction row (the concatenated crops of the batch: duplicates where crops
 *   overlap), pred (m, num_classes) f32;  mode 0: probs[inds] += pred, 1: probs[inds] = smooth * probs[inds] + (1 - smooth) * pred,
 *   2: probs[inds] = pred ('probs_last').  An indexed update ASSIGNS, so for a duplicated point one row of pred counts: the LAST one, as on
 *   the CPU (deterministic here; CUDA's index_put leaves it to the store order).  scratch_n: n ints. */
int cbl_cumulate_probs(int n, int num_classes, int m, const long long* inds, const float* pred, float smooth, int mode, float* probs,
                       int* scratch_n, void* stream);

/* boundary-IoU evaluation  pytorch/tool/test.py:392-417: get_boundary_mask (above, with get_plain) followed by
 *   intersectionAndUnion(pred[mask], label[mask], K, ignore)  util/common_util.py:25-37  for mask in (bound, plain), fused:
 *   pred (n) i64, labels (n) i64, neighbor_idx (n,k) -> hist (2,3,num_classes) u64 += [mask bound|plain][intersection|output|target]
 *   (caller pre-zeroes; union = output + target - intersection, :36) */
int cbl_boundary_iou(int n, int k, int num_classes, long long ignore_label, const long long* pred, const long long* labels,
                     const int* neighbor_idx, unsigned long long* hist, void* stream);

/* ------------------------------------------------------------------------------------------------
 * TF-side local aggregation over radius neighbourhoods (index == n0 selects the shadow / padding row)
 * ---------------------------------------------------------------------------------------------- */

/* a15  PseudoGrid = KPConv, depthwise  tensorflow/models/local_aggregation_operators.py:620-746 (math :681-728)
 *   query_points (n,3), support_points (n0,3), neighbors_indices (n,K) i32 (pad = n0), features (n0,C),
 *   kernel_points (KP,3) [KP <= 16; their generator create_kernel_points is absent from the reference: an input here],
 *   kernel_weights (KP,C), extent = KP_extent*radius/density_parameter (:664), influence 0 'constant' | 1 'linear' (:691-699),
 *   closest 0 'sum' | 1 'closest' (:705-708)  ->  out (n,C) = sum_kp kernel_weights[kp] * (w[kp,:] @ features[nbrs])  (before bn/act)
 *   The (KP x K)·(K x C) contraction runs on the matrix cores (v_mfma_f32_16x16x4_f32, exact f32). */
int cbl_kpconv_forward(int n, int n0, int K, int C, int KP, const float* query_points, const float* support_points, const int* neighbors_indices,
                       const float* features, const float* kernel_points, const float* kernel_weights, float extent, int influence, int closest,
                       float* out, void* stream);
/* same values; `order` (n ints, NULL = none) = processing sequence of the query points, see cbl_queryandgroup_ordered */
int cbl_kpconv_forward_ordered(int n, int n0, int K, int C, int KP, const float* query_points, const float* support_points, const int* neighbors_indices,
                               const float* features, const float* kernel_points, const float* kernel_weights, float extent, int influence, int closest,
                               const int* order, float* out, void* stream);
/* gradients w.r.t. features (n0,C) += and kernel_weights (KP,C) += (caller pre-zeroes; either may be NULL); K <= 64 */
int cbl_kpconv_backward(int n, int n0, int K, int C, int KP, const float* query_points, const float* support_points, const int* neighbors_indices,
                        const float* features, const float* kernel_points, const float* kernel_weights, float extent, int influence, int closest,
                        const float* grad_out, float* grad_features, float* grad_kernel_weights, void* stream);

/* the same gradients as a gather over the transposed table of neighbors_indices (cbl_neighbor_transpose with n targets = n0, pairs p = i*K + k):
 * no atomics, WRITTEN not accumulated (no pre-zeroing), deterministic.  C % 4 == 0 and 16-byte aligned rows (CBL_ERR_UNSUPPORTED otherwise: use
 * cbl_kpconv_backward); any K.  workspace (only for grad_kernel_weights): cbl_kpconv_backward_csr_workspace_bytes. */
size_t cbl_kpconv_backward_csr_workspace_bytes(int n0, int C, int KP);
int cbl_kpconv_backward_csr(int n, int n0, int K, int C, int KP, const float* query_points, const float* support_points, const float* features,
                            const float* kernel_points, const float* kernel_weights, float extent, int influence, int closest,
                            const float* grad_out, const int* order_dst, const int* inv_start, const int* inv_src,
                            float* grad_features, float* grad_kernel_weights, void* workspace, size_t workspace_bytes, void* stream);

/* a14  AdaptiveWeight  tensorflow/models/local_aggregation_operators.py:316-500 with the shipped options
 *   (config/s3dis/adapt.yaml:19-26: local_input_feature 'dp', fc_num 1, shared_channels 1, no softmax):
 *   w[p,k,:] = ((support[nbr]-query[p])/radius) @ fc_weight (3,C) + fc_bias (C);  out[p,:] = sum_k w[p,k,:]*features[nbr]  (/ nn[p] if reduction_mean)
 *   nn[p] = #{k: idx[p,k] < *padding_num} + 1e-5 with *padding_num = max over ALL of neighbors_indices (:466-470) — cbl_index_max computes it. */
int cbl_index_max(long long total, const int* idx, int* out_max, void* stream);
int cbl_adaptive_weight_forward(int n, int n0, int K, int C, const float* query_points, const float* support_points, const int* neighbors_indices,
                                const float* features, float radius, const float* fc_weight, const float* fc_bias, const int* padding_num,
                                int reduction_mean, float* out, void* stream);
/* same values; `order` (n ints, NULL = none) = processing sequence of the query points, see cbl_queryandgroup_ordered */
int cbl_adaptive_weight_forward_ordered(int n, int n0, int K, int C, const float* query_points, const float* support_points, const int* neighbors_indices,
                                        const float* features, float radius, const float* fc_weight, const float* fc_bias, const int* padding_num,
                                        int reduction_mean, const int* order, float* out, void* stream);
/* gradients: features (n0,C) +=, fc_weight (3,C) +=, fc_bias (C) +=  (caller pre-zeroes; any may be NULL) */
int cbl_adaptive_weight_backward(int n, int n0, int K, int C, const float* query_points, const float* support_points, const int* neighbors_indices,
                                 const float* features, float radius, const float* fc_weight, const float* fc_bias, const int* padding_num,
                                 int reduction_mean, const float* grad_out, float* grad_features, float* grad_fc_weight, float* grad_fc_bias, void* stream);

/* the same gradients as a gather over the transposed table of neighbors_indices (cbl_neighbor_transpose with n targets = n0, pairs p = i*K + k; shadow
 * neighbours are not in the table): no atomics, WRITTEN not accumulated (no pre-zeroing), deterministic — what tf.gather's gradient does for the reference
 * (local_aggregation_operators.py:360-484 under tf.gradients).  C % 4 == 0 and 16-byte aligned rows (CBL_ERR_UNSUPPORTED otherwise: use
 * cbl_adaptive_weight_backward). */
size_t cbl_adaptive_weight_backward_csr_workspace_bytes(int n, int n0, int C);
int cbl_adaptive_weight_backward_csr(int n, int n0, int K, int C, const float* query_points, const float* support_points, const int* neighbors_indices,
                                     const float* features, float radius, const float* fc_weight, const float* fc_bias, const int* padding_num,
                                     int reduction_mean, const float* grad_out, const int* order_dst, const int* inv_start, const int* inv_src,
                                     float* grad_features, float* grad_fc_weight, float* grad_fc_bias, void* workspace, size_t workspace_bytes, void* stream);

/* a14  PosPool  tensorflow/models/local_aggregation_operators.py:15-250 (shipped: config/s3dis/pospool.yaml:20-23 'sin_cos' + 'mean')
 *   out[p,c] = reduce_k geo[p,k,c/(C/mid)] * features[nbr(p,k),c]   (before pool_bn / activation / output_conv, :251-270)
 *   position_embedding: 0 'one' | 1 'xyz' | 2 'distance' | 3 'exp_-d' | 4 'direction_exp_-d' | 5 'direction_d' | 6 'sin_cos' |
 *                       7 'two_order' | 8 'three_order'   ('direction' alone cannot run in the reference: mid_fdim 1 vs a 3-vector, :93-96)
 *   reduction: 0 'sum' | 1 'mean' (nn[p] as in AdaptiveWeight, *padding_num from cbl_index_max, :236-242) | 2 'max' (:243-249)
 *   K <= 128.  CBL_ERR_UNSUPPORTED when C does not fit the embedding (the reference's reshape :229 fails there too). */
int cbl_pospool_forward(int n, int n0, int K, int C, const float* query_points, const float* support_points, const int* neighbors_indices,
                        const float* features, float radius, int position_embedding, int reduction, const int* padding_num,
                        float* out, void* stream);
/* gradient w.r.t. features (n0,C) += (caller pre-zeroes); 'max' shares the gradient among equal maxima like tf.reduce_max */
int cbl_pospool_backward(int n, int n0, int K, int C, const float* query_points, const float* support_points, const int* neighbors_indices,
                         const float* features, float radius, int position_embedding, int reduction, const int* padding_num,
                         const float* grad_out, float* grad_features, void* stream);
/* the same gradient for 'sum' / 'mean' as a gather over the transposed neighbour table of neighbors_indices (cbl_neighbor_transpose with n = n0: shadow
 * entries are not in it): grad_features (n0, C) is WRITTEN (no zero fill, no float atomics, deterministic).  C % 4 == 0; 'max' -> CBL_ERR_UNSUPPORTED
 * (use cbl_pospool_backward).  workspace: cbl_pospool_backward_csr_workspace_bytes(n). */
size_t cbl_pospool_backward_csr_workspace_bytes(int n);
int cbl_pospool_backward_csr(int n, int n0, int K, int C, const float* query_points, const float* support_points, const int* neighbors_indices,
                             float radius, int position_embedding, int reduction, const int* padding_num, const float* grad_out,
                             const int* order_dst, const int* inv_start, const int* inv_src, float* grad_features,
                             void* workspace, size_t workspace_bytes, void* stream);

/* a4  PointTransformerLayer  pytorch/model/blocks.py:31-44, the C-wide part without its (n,K,C) tensors (C = 32 or 64, G = C/8):
 *   p1 (n,K,3) = ReLU(BN(Linear(3,3)(p_j - p_i)))  [computed by the caller: narrow],  p_r = Linear(3,C)(p1) = p1 @ W3C^T + b3C  [never stored]
 * attn_w2:  w2 (n,K,G) = Linear(C,G)( ReLU( BN_C( x_k[idx] - x_q + p_r ) ) )       (:39 and the first half of linear_w, :25-27)
 *   training != 0: BatchNorm statistics over all n*K pairs (one extra pass), running stats / counter updated like nn.BatchNorm1d, batch mean /
 *   invstd left in save_mean / save_invstd (C); training == 0: save_mean / save_invstd are INPUTS (the running statistics).
 *   backward: grad_xq (n,C) written, grad_xk (n,C) += (caller pre-zeroes), grad_p1 (n,K,3), grad_W3C (C,3), grad_b3C, grad_bn_weight/bias (C),
 *   grad_Wa (G,C), grad_ba (G) written.
 * attn_agg: out (n,C) = sum_k (x_v[idx] + p_r) * a[.,.,c % G]                       (:42-43; K9/K10 with p_r on the fly)
 *   backward: grad_xv (n,C) += (caller pre-zeroes), grad_p1, grad_W3C, grad_b3C, grad_a (n,K,G) written.
 * Both Linear(3,C) gradients are partial (one from each use of p_r): the caller adds them.  workspace: cbl_attn_workspace_bytes. */
size_t cbl_attn_workspace_bytes(int C, int G);
int cbl_attn_w2_forward(int n, int K, int C, int G, const float* x_q, const float* x_k, const int* idx, const float* p1,
                        const float* W3C, const float* b3C, const float* bn_weight, const float* bn_bias, float eps, float momentum,
                        float* running_mean, float* running_var, long long* num_batches_tracked, int training,
                        const float* Wa, const float* ba, float* save_mean, float* save_invstd, float* w2,
                        void* workspace, size_t workspace_bytes, void* stream);
int cbl_attn_w2_backward(int n, int K, int C, int G, const float* x_q, const float* x_k, const int* idx, const float* p1,
                         const float* W3C, const float* b3C, const float* bn_weight, const float* bn_bias,
                         const float* save_mean, const float* save_invstd, const float* Wa, const float* grad_w2,
                         float* grad_xq, float* grad_xk, float* grad_p1, float* grad_W3C, float* grad_b3C,
                         float* grad_bn_weight, float* grad_bn_bias, float* grad_Wa, float* grad_ba,
                         void* workspace, size_t workspace_bytes, void* stream);
int cbl_attn_agg_forward(int n, int K, int C, int G, const float* x_v, const int* idx, const float* p1, const float* W3C, const float* b3C,
                         const float* a, float* out, void* stream);
int cbl_attn_agg_backward(int n, int K, int C, int G, const float* x_v, const int* idx, const float* p1, const float* W3C, const float* b3C,
                          const float* a, const float* grad_out, float* grad_xv, float* grad_p1, float* grad_W3C, float* grad_b3C, float* grad_a,
                          void* workspace, size_t workspace_bytes, void* stream);
/* The same pair with the softmax over the K neighbours (pytorch/model/blocks.py:41) inside: forward takes the logits (n, K, G) and writes the softmax
 * weights to `a` (kept for the backward pass), backward returns the gradient of the logits, a (da - sum over K of a da). */
int cbl_attn_agg_softmax_forward(int n, int K, int C, int G, const float* x_v, const int* idx, const float* p1, const float* W3C, const float* b3C,
                                 const float* logits, float* a, float* out, void* stream);
int cbl_attn_agg_softmax_backward(int n, int K, int C, int G, const float* x_v, const int* idx, const float* p1, const float* W3C, const float* b3C,
                                  const float* a, const float* grad_out, float* grad_xv, float* grad_p1, float* grad_W3C, float* grad_b3C,
                                  float* grad_logits, void* workspace, size_t workspace_bytes, void* stream);

/* the two backward passes with their scatters (grad_xk / grad_xv: sums over the pairs that list a row as neighbour — the index_select backward of
 * blocks.py:35-36,43 under autograd) as gathers over the transposed table of idx (cbl_neighbor_transpose with n targets = n): no atomics, grad_xk /
 * grad_xv WRITTEN not accumulated, run-to-run deterministic.  C = 32 / 64 (CBL_ERR_UNSUPPORTED for the wide stages).  softmax: as the two entries above. */
int cbl_attn_w2_backward_csr(int n, int K, int C, int G, const float* x_q, const float* x_k, const int* idx, const float* p1,
                             const float* W3C, const float* b3C, const float* bn_weight, const float* bn_bias,
                             const float* save_mean, const float* save_invstd, const float* Wa, const float* grad_w2,
                             const int* order, const int* inv_start, const int* inv_src,
                             float* grad_xq, float* grad_xk, float* grad_p1, float* grad_W3C, float* grad_b3C,
                             float* grad_bn_weight, float* grad_bn_bias, float* grad_Wa, float* grad_ba,
                             void* workspace, size_t workspace_bytes, void* stream);
int cbl_attn_agg_backward_csr(int n, int K, int C, int G, const float* x_v, const int* idx, const float* p1, const float* W3C, const float* b3C,
                              const float* a, const float* grad_out, const int* order, const int* inv_start, const int* inv_src,
                              float* grad_xv, float* grad_p1, float* grad_W3C, float* grad_b3C, float* grad_a,
                              void* workspace, size_t workspace_bytes, int softmax, void* stream);

/* a4  PointTransformerLayer  pytorch/model/blocks.py:31-44 as ONE pass structure (round 4): everything of the layer behind its three per-point
 *     Linear layers (blocks.py:33), train-mode BatchNorms, C = 32 | 64 (share_planes 8, G = C/8), K = 8 | 16 (the two full-resolution stages).
 *     Replaces, for those shapes, the call sequence queryandgroup -> linear_p -> subtraction -> linear_w -> softmax -> aggregation
 *     (pointops_api.cpp:12-23 functions 7-10 plus torch's Linear / BatchNorm1d / Softmax over (n,K,C) tensors).
 *   forward   xyz (n,3), x_q / x_k / x_v (n,C) = linear_q/k/v(x), idx (n,K) of the layer's self-search, order (n) = the search's cell order or NULL;
 *             parameters in the reference's layout: linear_p[0] Wp (3,3) bp (3), linear_p[1] gamma_p beta_p (3), linear_p[3] W3C (C,3) b3C (C),
 *             linear_w[0] gamma_c beta_c (C), linear_w[2] Wa (G,C) ba (G), linear_w[3] gamma_g beta_g (G), linear_w[5] Wb (G,G) bb (G);
 *             eps3 / momentum3: HOST arrays of 3 floats (BN_p, BN_c, BN_g); running_mean3 / running_var3 / num_batches3: HOST arrays of 3 device
 *             pointers (entries or the arrays may be NULL), updated as torch's train-mode BatchNorm1d does
 *             -> out (n,C);  kept for the backward pass: p_r, p0, p1 (n,K,3), w2, a (n,K,G), consts (cbl_pt_layer_consts_floats() floats).
 *   backward  + the transposed neighbour table of idx (cbl_neighbor_transpose: inv_start (n+1) by rank of `order`, inv_src), grad_out (n,C)
 *             -> g_xq, g_xk, g_xv (n,C) written (not accumulated, no atomics), and the 14 parameter gradients in the parameters' layouts.
 *   All sums in a fixed order: run-to-run deterministic.  CBL_ERR_UNSUPPORTED for other shapes (callers take the cbl_attn_* kernels above). */
size_t cbl_pt_layer_workspace_bytes(int n, int K, int C);
int cbl_pt_layer_consts_floats(void);
/* The same layer at the WIDE stages (C = 128 | 256 | 512, G = C / 8, K <= 64; training mode): everything behind the q / k / v projections as one call each
 * way — the p chain and the narrow (n, K, G) work by kernels of pt_layer.hip, the C-wide passes by the cbl_attn_* kernels above (pair values recomputed,
 * d x_k / d x_v by float atomics into buffers this call zeroes) — ~9 launches forward, ~14 backward instead of ~42 issued op by op.  Arguments as
 * cbl_pt_layer_forward / _backward; `a` receives the softmax weights, consts cbl_pt_layer_wide_consts_floats() floats, bnc_stats (2 C) the batch mean and
 * inverse standard deviation of BN_c (kept for the backward pass).  The table-based backward is not needed: no inv_start / inv_src. */
size_t cbl_pt_layer_wide_workspace_bytes(int n, int K, int C);
int cbl_pt_layer_wide_consts_floats(void);
int cbl_pt_layer_wide_forward(int n, int K, int C, const float* xyz, const float* x_q, const float* x_k, const float* x_v, const int* idx,
                              const float* Wp, const float* bp, const float* gamma_p, const float* beta_p, const float* W3C, const float* b3C,
                              const float* gamma_c, const float* beta_c, const float* Wa, const float* ba, const float* gamma_g, const float* beta_g,
                              const float* Wb, const float* bb, const float* eps3, const float* momentum3, float* const* running_mean3,
                              float* const* running_var3, long long* const* num_batches3, float* p_r, float* p0, float* p1, float* w2, float* a, float* out,
                              float* consts, float* bnc_stats, void* workspace, size_t workspace_bytes, void* stream);
int cbl_pt_layer_wide_backward(int n, int K, int C, const float* x_q, const float* x_k, const float* x_v, const int* idx, const float* gamma_p,
                               const float* W3C, const float* b3C, const float* gamma_c, const float* beta_c, const float* Wa, const float* gamma_g,
                               const float* Wb, const float* p_r, const float* p0, const float* p1, const float* w2, const float* a, const float* consts,
                               const float* bnc_stats, const float* grad_out, float* g_xq, float* g_xk, float* g_xv, float* g_Wp, float* g_bp,
                               float* g_gamma_p, float* g_beta_p, float* g_W3C, float* g_b3C, float* g_gamma_c, float* g_beta_c, float* g_Wa, float* g_ba,
                               float* g_gamma_g, float* g_beta_g, float* g_Wb, float* g_bb, void* workspace, size_t workspace_bytes, void* stream);

/* self-test of the numerical assumption the passes' agreeing ReLU masks rest on: one v_mfma_f32_16x16x4_f32 tile D = A (16,4) . B (4,16) + C (16,16)
 * (row-major device arrays) next to the k-ordered fmaf chain of the same tile; callers compare the two outputs bit for bit. */
int cbl_pt_layer_selftest_chain(const float* A, const float* B, const float* C, float* d_mfma, float* d_fma, void* stream);
int cbl_pt_layer_forward(int n, int K, int C, const float* xyz, const float* x_q, const float* x_k, const float* x_v, const int* idx, const int* order,
                         const float* Wp, const float* bp, const float* gamma_p, const float* beta_p, const float* W3C, const float* b3C,
                         const float* gamma_c, const float* beta_c, const float* Wa, const float* ba, const float* gamma_g, const float* beta_g,
                         const float* Wb, const float* bb, const float* eps3, const float* momentum3, float* const* running_mean3,
                         float* const* running_var3, long long* const* num_batches3, float* p_r, float* p0, float* p1, float* w2, float* a, float* out,
                         float* consts, void* workspace, size_t workspace_bytes, void* stream);
/* the layer in evaluation mode (nn.Module.eval(): the three BatchNorm1d normalise with their running statistics): no statistics passes, no buffer is
 * touched.  eps3: HOST array of 3 floats; running_mean3 / running_var3: HOST arrays of 3 device pointers (BN_p, BN_c, BN_g).  Scratch as the training entry. */
int cbl_pt_layer_forward_eval(int n, int K, int C, const float* xyz, const float* x_q, const float* x_k, const float* x_v, const int* idx, const int* order,
                              const float* Wp, const float* bp, const float* gamma_p, const float* beta_p, const float* W3C, const float* b3C,
                              const float* gamma_c, const float* beta_c, const float* Wa, const float* ba, const float* gamma_g, const float* beta_g,
                              const float* Wb, const float* bb, const float* eps3, const float* const* running_mean3, const float* const* running_var3,
                              float* p_r, float* p0, float* p1, float* w2, float* a, float* out, float* consts, void* workspace, size_t workspace_bytes,
                              void* stream);
int cbl_pt_layer_backward(int n, int K, int C, const float* x_q, const float* x_k, const float* x_v, const int* idx, const int* order,
                          const int* inv_start, const int* inv_src, const float* gamma_p, const float* W3C, const float* b3C, const float* gamma_c,
                          const float* Wa, const float* gamma_g, const float* Wb, const float* p_r, const float* p0, const float* p1, const float* w2,
                          const float* a, const float* consts, const float* grad_out, float* g_xq, float* g_xk, float* g_xv, float* g_Wp, float* g_bp,
                          float* g_gamma_p, float* g_beta_p, float* g_W3C, float* g_b3C, float* g_gamma_c, float* g_beta_c, float* g_Wa, float* g_ba,
                          float* g_gamma_g, float* g_beta_g, float* g_Wb, float* g_bb, void* workspace, size_t workspace_bytes, void* stream);

/* a4, dense part of the vector attention: nn.Linear over (n*K) rows with tiny widths — linear_p = Linear(3,3), Linear(3,C) and
 * linear_w = Linear(C,C/8), Linear(C/8,C/8)  pytorch/model/blocks.py:23-28,38-40 — as streaming kernels instead of library GEMMs.
 *   x (rows,cin), weight (cout,cin), bias (cout) or NULL -> y (rows,cout) = x @ weight^T + bias
 *   backward_input:  grad_x (rows,cin) = grad_y @ weight            backward_weight: grad_weight (cout,cin) = grad_y^T @ x, grad_bias (cout) = column sums of grad_y
 *   (grad_weight / grad_bias are written, not accumulated; grad_bias may be NULL; workspace: cbl_skinny_linear_workspace_bytes).
 *   cin*cout <= 4096 and cin+cout <= 200, else CBL_ERR_UNSUPPORTED. */
size_t cbl_skinny_linear_workspace_bytes(int cin, int cout);
int cbl_skinny_linear_forward(long long rows, int cin, int cout, const float* x, const float* weight, const float* bias, float* y, void* stream);
int cbl_skinny_linear_backward_input(long long rows, int cin, int cout, const float* grad_y, const float* weight, float* grad_x, void* stream);
int cbl_skinny_linear_backward_weight(long long rows, int cin, int cout, const float* x, const float* grad_y, float* grad_weight, float* grad_bias,
                                      void* workspace, size_t workspace_bytes, void* stream);

/* a4: the three per-point projections of the layer, x_q, x_k, x_v = linear_q(x), linear_k(x), linear_v(x)  pytorch/model/blocks.py:33 (nn.Linear(C, C) x 3,
 *     C = 32 | 64), one launch per direction.  weight3 / bias3 / y3 / grad_*3: HOST arrays of 3 device pointers (q, k, v).
 *   forward   y3[p] (rows, C) = x . weight3[p]^T + bias3[p]
 *   backward  grad_x = sum_p grad_y3[p] . weight3[p] (one pass, accumulated in the MFMA accumulators: no partial tensors, no adds);
 *             grad_weight3[p] (C, C) = grad_y3[p]^T . x; grad_bias3[p] (C) = column sums (array or entries may be NULL).  Written, not accumulated. */
size_t cbl_triple_linear_workspace_bytes(int C);
int cbl_triple_linear_forward(long long rows, int C, const float* x, const float* const* weight3, const float* const* bias3, float* const* y3, void* stream);
int cbl_triple_linear_backward(long long rows, int C, const float* x, const float* const* weight3, const float* const* grad_y3, float* grad_x,
                               float* const* grad_weight3, float* const* grad_bias3, void* workspace, size_t workspace_bytes, void* stream);

/* a4 / a5, dense part: train-mode nn.BatchNorm1d (+ ReLU) over (rows, C) activations, rows = n or n*K  pytorch/model/blocks.py:25-28,38-40,70,74,126-134
 *   y = [relu]((x - mean_batch) / sqrt(var_batch + eps) * weight + bias); running_mean / running_var updated in place like torch
 *   (momentum, unbiased variance; either may be NULL; *num_batches_tracked += 1 if not NULL); save_mean / save_invstd (C) are kept for the
 *   backward call.  weight / bias may be NULL.
 *   backward: grad_x (rows,C), grad_weight (C), grad_bias (C) (written, not accumulated; the last two may be NULL); `relu` masks grad_y
 *   where the forward output was 0.  C <= 1024 (C % 4 != 0: C <= 256).  workspace: cbl_bn_rows_workspace_bytes. */
size_t cbl_bn_rows_workspace_bytes(long long rows, int C);
int cbl_bn_rows_forward(long long rows, int C, const float* x, const float* weight, const float* bias, float eps, float momentum,
                        float* running_mean, float* running_var, long long* num_batches_tracked, int relu, float* save_mean, float* save_invstd,
                        float* y, void* workspace, size_t workspace_bytes, void* stream);
int cbl_bn_rows_backward(long long rows, int C, const float* x, const float* grad_y, const float* weight, const float* bias,
                         const float* save_mean, const float* save_invstd, int relu, float* grad_x, float* grad_weight, float* grad_bias,
                         void* workspace, size_t workspace_bytes, void* stream);
/* the tail of a residual block as ONE BatchNorm call  pytorch/model/blocks.py:130-133 (x = bn3(linear3(x)); x += identity; x = relu(x))
 *   y = [relu](bn(x) + residual); residual (rows,C) or NULL (then exactly cbl_bn_rows_forward).
 *   backward: the mask is (bn(x) + residual > 0) recomputed from x and residual; grad_residual (rows,C) or NULL receives the masked grad_y
 *   (the gradient of the skip connection), grad_x / grad_weight / grad_bias as above. */
int cbl_bn_rows_forward_residual(long long rows, int C, const float* x, const float* residual, const float* weight, const float* bias, float eps, float momentum,
                                 float* running_mean, float* running_var, long long* num_batches_tracked, int relu, float* save_mean, float* save_invstd,
                                 float* y, void* workspace, size_t workspace_bytes, void* stream);
int cbl_bn_rows_backward_residual(long long rows, int C, const float* x, const float* residual, const float* grad_y, const float* weight, const float* bias,
                                  const float* save_mean, const float* save_invstd, int relu, float* grad_x, float* grad_residual, float* grad_weight,
                                  float* grad_bias, void* workspace, size_t workspace_bytes, void* stream);

/* the criterion's cross entropy  pytorch/model/pointtransformer_seg.py:20-22 (nn.CrossEntropyLoss(ignore_index), reduction 'mean')
 *   logits (n,k) f32, k <= 64; target (n) i64; points with target == ignore_index do not count; any OTHER target outside [0,k) makes the loss NaN (the library
 *   device-asserts on it; zero gradient rows for those points).
 *   forward: loss (1) = sum_i (logsumexp(logits[i]) - logits[i, target[i]]) / count; stats (2) = {sum, count}, kept for the backward call.
 *   backward: grad_logits (n,k) = (softmax(logits[i]) - onehot(target[i])) * grad_loss[0] / count, 0 for points that do not count (written, not accumulated).
 *   Deterministic (per-workgroup partial sums combined in fp64 in a fixed order).  workspace: cbl_cross_entropy_workspace_bytes. */
size_t cbl_cross_entropy_workspace_bytes(long long n);
int cbl_cross_entropy_forward(long long n, int k, const float* logits, const long long* target, long long ignore_index, float* loss, float* stats,
                              void* workspace, size_t workspace_bytes, void* stream);
int cbl_cross_entropy_backward(long long n, int k, const float* logits, const long long* target, long long ignore_index, const float* stats,
                               const float* grad_loss, float* grad_logits, void* stream);

/* ind_max_pool / ind_closest_pool  tensorflow/models/basic_operators.py:155-172 / :175-192
 *   x (n1,d), inds (n2,k) i32 (pad = n1) -> out (n2,d): max over the row's entries (shadow row = column-wise min of x; scratch_d (d) u32)
 *   / the entry of the FIRST column (shadow row = 0) */
int cbl_ind_max_pool(int n1, int n2, int k, int d, const float* x, const int* inds, unsigned* scratch_d, float* out, void* stream);
int cbl_ind_closest_pool(int n1, int n2, int k, int d, const float* x, const int* inds, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * TF-side CPU ops of the reference (stacked clouds; here described by cumulative END offsets like the
 * pytorch side — the TF ops take per-cloud lengths, the host mirror converts)
 * ---------------------------------------------------------------------------------------------- */

/* N1 / N3  batch_grid_subsampling  tensorflow/ops/tf_custom_ops/tf_subsampling/grid_subsampling/grid_subsampling.cpp:114-161
 *          (TF op BatchGridSubsampling, tf_batch_subsampling.cpp:8-20), and with features / labels the dataset
 *          preprocessing flavour cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-106
 *          (grid_subsampling.compute(points, features=, classes=, sampleDl=), wrapper.cpp:70-76).
 *   points (n,3), offset (b) [b <= 65535], dl; optional features (n,fdim) f32 -> per-voxel mean, labels (n,ldim) i32 -> per-voxel
 *   majority (ties: smallest label; the reference's tie winner is hash-map order)
 *   -> out_points (>= n rows capacity; first *out_total valid, voxels in ascending key order cloud by cloud), out_features,
 *      out_labels, out_lengths (b) voxels per cloud, out_total (1).  Barycentres are bit-identical to the reference
 *      (same accumulation order); only the ORDER of the voxels differs (the reference emits unordered_map iteration order). */
size_t cbl_grid_subsampling_workspace_bytes(int b, int n);
int cbl_grid_subsampling(int b, int n, const float* points, const int* offset, float dl,
                         int fdim, const float* features, int ldim, const int* labels,
                         float* out_points, float* out_features, int* out_labels, int* out_lengths, int* out_total,
                         void* workspace, size_t workspace_bytes, void* stream);

/* N2  batch_nanoflann_neighbors  tensorflow/ops/tf_custom_ops/tf_neighbors/neighbors/neighbors.cpp:213-336 (TF op
 *     BatchOrderedNeighbors, tf_batch_neighbors.cpp:8-30) fused with the callers' crop to neighborhood_limits
 *     (tensorflow/datasets/base.py:756-765).
 *   queries (nq,3) / q_offset (b), supports (ns,3) / s_offset (b), radius, limit (<= 64)
 *   -> out (nq,limit) i32: the supports with d2 < radius^2 (strict), ascending by (d2, index), global row ids, padded with ns;
 *      counts (nq, may be NULL) true number inside the ball; max_count (1) = the reference's output width before the crop. */
/* The reference's NON-batch TF ops — GridSubsampling(points, dl) (tf_subsampling.cpp:8-20 over grid_subsampling.cpp:6-112) and
 * OrderedNeighbors(queries, supports, radius) (tf_neighbors.cpp:8-62 over neighbors.cpp:58-208) — are the batch entries with b = 1
 * (offset = {n}): same barycentres / neighbour order.  Host mirrors: tf_ops.tf_grid_subsampling, tf_ops.tf_ordered_neighbors. */
size_t cbl_radius_neighbors_workspace_bytes(int b, int ns);
int cbl_radius_neighbors(int b, int nq, int ns, const float* queries, const float* supports, const int* q_offset, const int* s_offset,
                         float radius, int limit, int* out, int* counts, int* max_count, void* workspace, size_t workspace_bytes, void* stream);
/* the same search when `workspace` still holds the grid an earlier cbl_radius_neighbors / _reuse call on this stream built for the SAME supports,
 * s_offset and radius (grid_is_built != 0: the 5-launch grid build is skipped; 0: as cbl_radius_neighbors).  The pyramid builder
 * (tensorflow/datasets/base.py:795-812) searches every layer's points two or three times with one radius: 13 searches, 5 distinct grids.
 * A call with nq = 0 builds nothing. */
int cbl_radius_neighbors_reuse(int b, int nq, int ns, const float* queries, const float* supports, const int* q_offset, const int* s_offset,
                               float radius, int limit, int* out, int* counts, int* max_count, void* workspace, size_t workspace_bytes,
                               int grid_is_built, void* stream);

/* One layer of the pyramid builder tf_segmentation_inputs_radius (tensorflow/datasets/base.py:795-812; last layer :815-820) as one host call:
 *   neighbors = N2(points, points, r) cropped to limit; (pool_points, pool_lengths) = N1(points, sample_dl); pools = N2(pool_points, points, r);
 *   upsamples = N2(points, pool_points, 2 r) — the same kernels in the same order as the separate entries (bit-identical tables), issued by native
 *   host code: no interpreter between the ~30 launches, and the layer's one data-dependent host wait (the sub-sampled point count, returned in
 *   *host_pool_points — pinned memory makes the copy asynchronous) inside the call, so a loader thread can run it beside the training thread.
 *   lengths (b) are per-cloud point counts (the TF side's convention).  sample_dl = 0: last layer, only `neighbors` (the pool / upsample / next-grid
 *   arguments may be NULL).  grid_ws holds (grid_is_built != 0) or receives the grid of (points, r); next_grid_ws receives the grid of
 *   (pool_points, 2 r) — the next layer's grid_ws with grid_is_built = 1; both cbl_radius_neighbors_workspace_bytes(b, n) bytes.
 *   pool_points / pools have capacity n rows; max_counts (3, device) = the three searches' largest neighbourhoods (neighbors, pools, upsamples). */
size_t cbl_pyramid_layer_workspace_bytes(int b, int n);
int cbl_pyramid_layer(int b, int n, const float* points, const int* lengths, float radius, float sample_dl, int limit,
                      void* grid_ws, size_t grid_ws_bytes, int grid_is_built,
                      int* neighbors, float* pool_points, int* pool_lengths, int* pools, int* upsamples,
                      void* next_grid_ws, size_t next_grid_ws_bytes, int* max_counts, int* host_pool_points,
                      void* workspace, size_t workspace_bytes, void* stream);

/* The whole pyramid of tf_segmentation_inputs_radius (base.py:784-820) in one call: cbl_pyramid_layer for every layer, all outputs at the capacity of
 * layer 0 (n rows), the sizes handed on through host_sizes (num_layers ints on the host, [0] = n; valid when the call returns).  limits (host, num_layers);
 * arrays indexed by layer: grid_ws[l] (cbl_radius_neighbors_workspace_bytes(b, n) bytes each), neighbors[l] (n, limits[l]); for l < num_layers - 1:
 * pool_points[l] (n, 3), pool_lengths[l] (b), pools[l] (n, limits[l]), upsamples[l] (n, limits[l]); max_counts (3 num_layers, device);
 * workspace cbl_pyramid_layer_workspace_bytes(b, n). */
int cbl_pyramid(int b, int n, const float* points, const int* lengths, float radius0, float dl0, int num_layers, const int* limits,
                void* const* grid_ws, size_t grid_ws_bytes, int* const* neighbors, float* const* pool_points, int* const* pool_lengths,
                int* const* pools, int* const* upsamples, int* max_counts, int* host_sizes, void* workspace, size_t workspace_bytes, void* stream);

/* The ConvNet's per-scene work behind the pyramid as ONE host call (no counterpart as a single function in the reference: there these are TF graph ops
 * issued by the TF runtime — local_aggregation_operators.py:360-484 under tf.gradients, heads/head.py:25-49 and :462-807): for every layer
 * AdaptiveWeight forward + backward ('mean' reduction, gradients as gathers over the layer's transposed neighbour table), the scene labels of the
 * layer (layer 0: point_labels; layer l: arg-max of the label histogram over `pools`, indices into layer l-1, pad = its point count) and the
 * contrast head ('softnn', 'l2', sample 'label') forward + gradient w.r.t. `latent` (loss upstream gradient 1).  The same kernels, in the same order,
 * as cbl_index_max / cbl_adaptive_weight_forward / cbl_neighbor_transpose / cbl_adaptive_weight_backward_csr / cbl_tf_scene_label /
 * cbl_label_argmax / cbl_contrast_pairs_forward_samples / cbl_contrast_pairs_backward called one by one; nothing waits for the device.
 * All pointers are device pointers; outputs are written (no pre-zeroing).  C % 4 == 0, d in {4,8,16,32,64}, K <= 65. */
typedef struct CblConvnetLayer {
    int n, K, C, Kp, d;            /* points of the layer; width of `neighbors`; AdaptiveWeight width; width of `pools` (0 at layer 0); latent width */
    float radius;                  /* AdaptiveWeight's radius of this layer */
    const float* points;           /* (n,3) */
    const int* neighbors;          /* (n,K) radius neighbours incl. the self column, padded with n */
    const float* features;         /* (n,C) */
    const float* fc_weight;        /* (3,C) */
    const float* fc_bias;          /* (C) */
    const float* grad_out;         /* (n,C) upstream gradient of the aggregation's output */
    const float* latent;           /* (n,d) features of the contrast head */
    const int* pools;              /* (n,Kp) pooling neighbours into layer l-1 (NULL at layer 0) */
    float* aw_out;                 /* (n,C) */
    float* grad_features;          /* (n,C) */
    float* grad_fc_weight;         /* (3,C) */
    float* grad_fc_bias;           /* (C) */
    float* cbl_loss;               /* (1) */
    int* cbl_mask;                 /* (n) 1 = the point has positive and negative neighbours */
    float* grad_latent;            /* (n,d) */
    int* labels;                   /* (n) hard scene labels of this layer */
} CblConvnetLayer;
size_t cbl_convnet_step_workspace_bytes(int nlayers, const CblConvnetLayer* layers, int num_classes);
int cbl_convnet_step(int nlayers, const CblConvnetLayer* layers, const long long* point_labels, int num_classes, float temperature, float weight,
                     void* workspace, size_t workspace_bytes, void* stream);

/* N4  cpp_knn_batch_omp  tensorflow/ops/nearest_neighbors/knn_.cxx:104-135: dense batch (B,N,3) x (B,M,3) -> (B,M,K) int64 LOCAL indices.
 *     = cbl_knnquery on the flattened batch (offset = N, 2N, ...) followed by this conversion of the global int32 rows. */
int cbl_knn_indices_to_local(int B, int M, int K, int N, const int* idx, long long* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Dataloader stage of the pytorch side (numpy on CPU workers in the reference)
 * ---------------------------------------------------------------------------------------------- */

/* a13  voxelize + fnv_hash_vec  pytorch/util/voxelize.py:4-16, :38-56
 *   coord (n,3) f32 or f64 (is_f64), already shifted to >= 0 (data_util.py:52-53), voxel_size
 *   -> keys_sorted (n) u64 = sorted FNV keys, idx_sort (n) i32 = argsort(key) [stable: ascending index inside a voxel; the reference's
 *      quicksort order inside a voxel is unspecified], start / count (capacity n) per voxel in key order, num_voxels (1).
 *   mode 1 of the reference returns (idx_sort, count); mode 0 picks idx_sort[start + rand % count] (host mirror). */
size_t cbl_voxelize_workspace_bytes(int n);
int cbl_voxelize(int n, int is_f64, const void* coord, double voxel_size, unsigned long long* keys_sorted, int* idx_sort,
                 int* start, int* count, int* num_voxels, void* workspace, size_t workspace_bytes, void* stream);

/* crop to the voxel_max points nearest to a centre  pytorch/util/data_util.py:62-64
 *   -> order (n) i32 = argsort of squared distance to coord[center] (stable); workspace as cbl_voxelize */
int cbl_crop_order(int n, int is_f64, const void* coord, int center, int* order, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CBL_AMD_H */
