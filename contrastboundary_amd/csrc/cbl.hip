// F5 / a7 / a9: Contrastive Boundary Learning head — sub-scene labels, positive/negative pair mining,
// soft-nearest-neighbour loss (forward + analytic backward), boundary masks.
// Replaces the torch-op chains of
//   get_subscene_label / get_subscene_features   /root/reference/pytorch/model/basic_operators.py:9-50
//   ContrastHead.point_contrast                  /root/reference/pytorch/model/heads.py:185-246
//       (posmask_cnt :145-149, dist_l2 :116-119, contrast_softnn :151-165)
//   get_boundary_mask                            /root/reference/pytorch/model/basic_operators.py:69-97
//
// MI355X mapping: the reference materialises neighbor_label (m,K-1,ncls), neighbor_feature (m,K-1,d), boolean
// masks, compacts rows with a host sync (torch.any, heads.py:222) and runs ~15 elementwise kernels.  Here one
// G-lane group owns one point (G = 16/32/64 >= K-1): lane j gathers neighbour j's whole feature row with 16 B
// loads (rows are 128 B at d=32: whole cache lines), reduces |f_i - f_j|^2 in registers, and the max / sums of the
// soft-NN loss are cross-lane reductions inside the group.  Nothing but the per-point loss (4 B) is written; rows
// without both a positive and a negative neighbour are masked, not compacted, so there is no host sync.
// Backward recomputes the same quantities and scatters with fp32 L2 atomics, one coalesced 4*d-byte row per
// (point, neighbour) — the same access pattern autograd's index_select backward has in the reference.
#include "cbl_common.h"
#include "wave_ops.h"

namespace {

// ---- a7: soft label = mean one-hot of the kr nearest stage-0 points ------------------------------------
// one wave per query; lanes sweep the kr neighbours, class counts by ballot + popcount
// n_valid: neighbour ids >= n_valid (the TF radius search's padding) and negative labels count for nothing; by_valid: divide by the
// number of valid neighbours + 1e-12 (get_neighbor_summary 'soft', tensorflow/models/heads/head.py:117-124) instead of by kr
__global__ __launch_bounds__(256) void subscene_label_kernel(int m, int kr, int ncls, const long long* __restrict__ target,
                                                             const int* __restrict__ nidx, int n_valid, int by_valid, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int q = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (q >= m) return;
    int mycount = 0, nvalid = 0;                        // lane c accumulates the count of class c (ncls <= 64)
    for (int base = 0; base < kr; base += 64) {
        const int j = base + lane;
        const int id = (j < kr) ? nidx[(size_t)q * kr + j] : -1;
        const int lab = (id >= 0 && id < n_valid) ? (int)target[id] : -1;
        nvalid += __popcll(__ballot(lab >= 0));
        for (int c = 0; c < ncls; c++) {
            const int cnt = __popcll(__ballot(lab == c));
            if (lane == c) mycount += cnt;
        }
    }
    if (lane < ncls) out[(size_t)q * ncls + lane] = by_valid ? (float)mycount / ((float)nvalid + 1e-12f)
                                                             : (float)mycount / (float)kr;              // x.float().mean(-2), :41
}

// ---- argmax over classes, first maximal index (torch.argmax) -------------------------------------------
__global__ __launch_bounds__(256) void label_argmax_kernel(int m, int ncls, const float* __restrict__ labels, int* __restrict__ amax)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const float* row = labels + (size_t)i * ncls;
    float best = row[0]; int bi = 0;
    for (int c = 1; c < ncls; c++) { const float v = row[c]; if (v > best) { best = v; bi = c; } }
    amax[i] = bi;
}

// per-group state shared by forward and backward
template <int G, int DV>
struct ContrastRow {
    bool nb;          // this lane holds a real neighbour (j < ns)
    bool pos;         // neighbour has the centre's label
    bool valid;       // the point has both positive and negative neighbours (group-uniform)
    float dist, e, P, A;
    float diff[DV * 4];
    int nbr;
};

// n_valid / tf_variant select the TF flavour of the head (tensorflow/models/heads/head.py:462-807, sample 'label', contrast
// 'softnn'): neighbour ids >= n_valid are the radius search's shadow padding (and negative hard labels are ignored points) and
// take no part (valid mask, :540-545, :626-640); dist = sqrt(max(sum, 1e-12)) (:184-185) instead of sqrt(sum + 1e-12); the
// max-shift runs over every column, masked or not (:752).  A point counts if it has >= 1 valid positive and >= 1 valid negative.
template <int G, int DV>
__device__ __forceinline__ void contrast_row(ContrastRow<G, DV>& r, int i, int gl, int nsample, int d, const float* __restrict__ feat,
                                             const int* __restrict__ amax, const int* __restrict__ nidx, float inv_temperature,
                                             int n_valid, int flags, float kl_thr)
{
    // flags: bit 0 = TF flavour; bit 1 = `amax` points at int64 labels (the reference's torch.long targets) read through their low words,
    // which saves the caller a conversion pass (class ids and ignore labels fit 32 bits); bits 8..15 = ncls > 0: `amax` points at SOFT
    // labels (n_valid x ncls floats) and a neighbour is a positive when KL(p_centre || p_neighbour) < kl_thr (sample 'labelkl<thr>')
    const int tf_variant = flags & 1, ls = 1 + ((flags >> 1) & 1), ncls = (flags >> 8) & 0xff;
    const int ns = nsample - 1;                                     // self column dropped, heads.py:195-196 / head.py:560
    const bool col = gl < ns;
    const int raw = nidx[(size_t)i * nsample + 1 + (col ? gl : 0)];
    const bool real = raw >= 0 && raw < n_valid;
    r.nbr = real ? raw : 0;
    if (ncls) {
        // collect_labels head.py:498-511 with calc_dist 'kl' :189-191: sum_c xlogy(p_i[c], p_i[c] / max(p_j[c], 1e-12)), classes in ascending
        // order; a shadow neighbour gathers the zero row (shadow_fn=0); only the shadow mask restricts the pairs (no ignored labels: mask_c None)
        const float* __restrict__ soft = reinterpret_cast<const float*>(amax);
        float kl = 0.f;
        for (int c = 0; c < ncls; c++) {
            const float pi = soft[(size_t)i * ncls + c], pj = real ? soft[(size_t)r.nbr * ncls + c] : 0.f;
            if (pi > 0.f) kl += pi * logf(pi / fmaxf(pj, 1e-12f));
        }
        r.nb = col && real;
        r.pos = r.nb && (kl < kl_thr);                              // :511
    } else {
        const int my = amax[(size_t)i * ls], nl = amax[(size_t)r.nbr * ls];
        r.nb = col && real && (!tf_variant || (my >= 0 && nl >= 0));    // takes part in the sums
        r.pos = r.nb && (nl == my);                                 // posmask_cnt :145-149 / head.py:538
    }
    const int cnt = group_sum_i<G>(r.pos ? 1 : 0);
    const int nvalid = group_sum_i<G>(r.nb ? 1 : 0);
    r.valid = cnt > 0 && cnt < nvalid;                              // :212-213 / solve_samples_mask head.py:621-640
    // a point without both kinds of neighbours contributes neither loss nor gradient (:212-213, :233): its group stops before the
    // feature gather — the labels decide, and only the boundary points (a fraction of the scene) pay for the 4*d*ns-byte gather
    if (!r.valid) { r.dist = 1.f; r.e = 0.f; r.P = 0.f; r.A = 1.f; return; }
    const float4* fi = reinterpret_cast<const float4*>(feat + (size_t)i * d);
    const float4* fj = reinterpret_cast<const float4*>(feat + (size_t)r.nbr * d);
    float acc = 0.f;
#pragma unroll
    for (int v = 0; v < DV; v++) {
        const float4 a = fi[v], b = fj[v];
        r.diff[4 * v + 0] = a.x - b.x; r.diff[4 * v + 1] = a.y - b.y; r.diff[4 * v + 2] = a.z - b.z; r.diff[4 * v + 3] = a.w - b.w;
#pragma unroll
        for (int k = 0; k < 4; k++) acc += r.diff[4 * v + k] * r.diff[4 * v + k];
    }
    r.dist = tf_variant ? sqrtf(fmaxf(acc, 1e-12f)) : sqrtf(acc + 1e-12f);   // head.py:184-185 / dist_l2 heads.py:116-119
    // shadow columns of the TF flavour gather a zero feature row and DO enter the max-shift (head.py:752)
    float shadow_d = 0.f;
    if (tf_variant && col && !real) {
        float a2 = 0.f;
#pragma unroll
        for (int v = 0; v < DV; v++) { const float4 a = fi[v]; a2 += a.x * a.x; a2 += a.y * a.y; a2 += a.z * a.z; a2 += a.w * a.w; }
        shadow_d = sqrtf(fmaxf(a2, 1e-12f));
    }
    float neg = r.nb ? -r.dist : ((tf_variant && col) ? (real ? -r.dist : -shadow_d) : -INFINITY);
    const float mx = group_max<G>(neg);                             // :153
    // pytorch: shift, then / T (:153-155); TF: / T, then shift (head.py:750-752) — the same value up to rounding
    neg = (neg - mx) * inv_temperature;
    r.e = r.nb ? expf(neg) : 0.f;
    r.P = group_sum<G>(r.pos ? r.e : 0.f);
    r.A = group_sum<G>(r.e);
}

template <int G, int DV>
__global__ __launch_bounds__(256) void contrast_fwd_kernel(int m, int nsample, const float* __restrict__ feat, const int* __restrict__ amax,
                                                           const int* __restrict__ nidx, float inv_temperature, int n_valid, int tf_variant, float kl_thr,
                                                           float* __restrict__ per_point, int* __restrict__ point_mask)
{
    const int t = (blockIdx.x * 256 + threadIdx.x) / G;
    const int gl = threadIdx.x & (G - 1);
    const int i = t < m ? t : m - 1;
    ContrastRow<G, DV> r;
    contrast_row<G, DV>(r, i, gl, nsample, DV * 4, feat, amax, nidx, inv_temperature, n_valid, tf_variant, kl_thr);
    if (t < m && gl == 0) {
        per_point[t] = r.valid ? -logf(r.P / r.A + 1e-12f) : 0.f;  // contrast_softnn :161-163
        point_mask[t] = r.valid ? 1 : 0;
    }
}

// deterministic reduction: stats[0] = sum of per-point losses, stats[1] = #valid points, loss = w * mean
__global__ __launch_bounds__(1024) void contrast_finalize_kernel(int m, float weight, const float* __restrict__ per_point,
                                                                 const int* __restrict__ point_mask, float* __restrict__ stats, float* __restrict__ loss)
{
    __shared__ float ssum[16]; __shared__ float scnt[16];
    float s = 0.f, c = 0.f;
    // one workgroup, so the pass is a chain of dependent round trips: 16-byte loads, all of a thread's loads in flight at once
    const int m4 = ((reinterpret_cast<size_t>(per_point) | reinterpret_cast<size_t>(point_mask)) & 15) ? 0 : (m >> 2);
    const float4* pp4 = reinterpret_cast<const float4*>(per_point);
    const int4* pm4 = reinterpret_cast<const int4*>(point_mask);
#pragma unroll 4
    for (int i = threadIdx.x; i < m4; i += 1024) {
        const float4 a = pp4[i]; const int4 b = pm4[i];
        s += (a.x + a.y) + (a.z + a.w); c += (float)(b.x + b.y + b.z + b.w);
    }
    for (int i = 4 * m4 + threadIdx.x; i < m; i += 1024) { s += per_point[i]; c += (float)point_mask[i]; }
    for (int k = 32; k >= 1; k >>= 1) { s += __shfl_xor(s, k); c += __shfl_xor(c, k); }
    if ((threadIdx.x & 63) == 0) { ssum[threadIdx.x >> 6] = s; scnt[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float S = 0.f, C = 0.f;
        for (int w = 0; w < 16; w++) { S += ssum[w]; C += scnt[w]; }
        stats[0] = S; stats[1] = C;
        loss[0] = C > 0.f ? (S / C) * weight : 0.f;                 // torch.mean(loss) * float(w), :241-243; 0 if no boundary point (:233)
    }
}

// Backward.  Phase 1 (lane = neighbour, as in the forward): the scalar coefficient of every pair,
//   coef_j = g*w/count * e_j (pos_j*A - P) / (T A^2 (P/A + eps)) / dist_j,   d loss/d f_i += coef_j (f_i - f_j),   d loss/d f_j -= same.
// Phase 2 (lane = channel): the wave walks its pairs R = 64/d at a time; each pair is ONE coalesced 4*d-byte row:
// re-load f_j[c], scatter -coef_j*(f_i[c]-f_j[c]) with a row of L2 atomics, accumulate the centre's own gradient in
// registers (one atomic row per point at the end).  (Scattering from the lane=neighbour layout would issue one 4-byte
// atomic per lane per channel, all to different rows: measured 10x slower.)
template <int G, int DV>
__global__ __launch_bounds__(256) void contrast_bwd_kernel(int m, int nsample, const float* __restrict__ feat, const int* __restrict__ amax,
                                                           const int* __restrict__ nidx, float inv_temperature, float weight,
                                                           int n_valid, int tf_variant, float kl_thr,
                                                           const float* __restrict__ stats, const float* __restrict__ grad_loss,
                                                           float* __restrict__ grad_feat,
                                                           float* __restrict__ per_point, int* __restrict__ point_mask)   // non-null: fused forward
{
    // Fused forward + gradient (per_point != nullptr): the point's loss term is written as in contrast_fwd_kernel and the gradient is
    // accumulated WITHOUT its global factor g*w/count (count = number of qualifying points is only known after the whole launch);
    // cbl_contrast_grad_scale applies the factor in the backward pass.  One gather of the neighbour rows instead of two.
    const bool fused = per_point != nullptr;
    constexpr int D = DV * 4;                                       // channels
    constexpr int R = 64 / D;                                       // pairs per step in phase 2 (D <= 64)
    const int lane = threadIdx.x & 63;
    const int t = (blockIdx.x * 256 + threadIdx.x) / G;
    const int gl = threadIdx.x & (G - 1);
    const int i = t < m ? t : m - 1;
    const int ns = nsample - 1;
    const float count = fused ? 1.f : stats[1];
    if (!(count > 0.f)) return;                                     // uniform: the loss was the constant 0
    float coef = 0.f; int nbr;
    {
        ContrastRow<G, DV> r;
        contrast_row<G, DV>(r, i, gl, nsample, D, feat, amax, nidx, inv_temperature, n_valid, tf_variant, kl_thr);
        nbr = r.nbr;
        if (fused && t < m && gl == 0) {
            per_point[t] = r.valid ? -logf(r.P / r.A + 1e-12f) : 0.f;   // contrast_softnn :161-163
            point_mask[t] = r.valid ? 1 : 0;
        }
        if (t < m && r.valid && r.nb) {
            const float scale = fused ? 1.f : grad_loss[0] * weight / count;
            const float ratio = r.P / r.A;
            // ((pos ? A : 0) - P) / A^2 as two quotients by A: A^2 underflows in fp32 once A < 1e-19 (see cbl_pairs.hip)
            coef = scale * (r.e / r.A) * (((r.pos ? r.A : 0.f) - r.P) / r.A) * inv_temperature / (ratio + 1e-12f) / r.dist;
            if ((tf_variant & 1) && r.dist <= 1e-6f) coef = 0.f;     // sqrt(max(s, 1e-12)): flat below the clamp
        }
    }
    // phase 2: groups of this wave one after the other (wave-uniform loop), lanes = (pair slot, channel).
    // (Handing the differences of phase 1 over through LDS instead of re-gathering f_j was measured SLOWER, 100 vs 93 us: the
    // kernel is bound by the device-scope atomics, ~225 G float atomics/s, not by the second gather, which hits L2.)
    if (__ballot(coef != 0.f) == 0ull) return;                      // no pair of this wave carries a gradient
    const int ch = lane % D, slot = lane / D;
    for (int g = 0; g < 64 / G; g++) {
        const int pt = ((blockIdx.x * 256 + (threadIdx.x & ~63)) / G) + g;          // point of group g (wave-uniform)
        if (pt >= m) break;
        const float fi = feat[(size_t)pt * D + ch];
        float acc = 0.f;
        // PB steps at a time: all their neighbour rows are requested before the first atomic is issued (a load cannot be moved
        // above an atomic by the compiler, so the plain loop paid one L2 round trip per step: ns/R of them in a row)
        constexpr int PB = 16;
        for (int b0 = 0; b0 < ns; b0 += PB * R) {
            float fj[PB];
#pragma unroll
            for (int u = 0; u < PB; u++) {
                const int j = b0 + u * R + slot;
                const int src = g * G + (j < ns ? j : 0);
                const int nj = __shfl(nbr, src);
                fj[u] = (b0 + u * R < ns) ? feat[(size_t)nj * D + ch] : 0.f;      // nbr is a valid row for every lane
            }
#pragma unroll
            for (int u = 0; u < PB; u++) {
                const int j = b0 + u * R + slot;
                const int src = g * G + (j < ns ? j : 0);
                const float cj = __shfl(coef, src);                 // 0 for masked-out points / padding lanes
                const int nj = __shfl(nbr, src);
                if (j < ns && cj != 0.f) {
                    const float gch = cj * (fi - fj[u]);
                    unsafeAtomicAdd(grad_feat + (size_t)nj * D + ch, -gch);
                    acc += gch;
                }
            }
        }
#pragma unroll
        for (int sft = D; sft < 64; sft <<= 1) acc += __shfl_xor(acc, sft);         // combine the R pair slots
        if (slot == 0 && acc != 0.f) unsafeAtomicAdd(grad_feat + (size_t)pt * D + ch, acc);
    }
}

// ---- a9: boundary / plain masks from neighbour labels ---------------------------------------------------
__global__ __launch_bounds__(256) void boundary_mask_kernel(int n, int k, const long long* __restrict__ labels, const int* __restrict__ nidx,
                                                            unsigned char* __restrict__ bound, unsigned char* __restrict__ plain, int* __restrict__ cnt)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long me = labels[i];
    int neq = 0; bool all_eq = true;
    for (int j = 0; j < k; j++) {
        const long long nl = labels[nidx[(size_t)i * k + j]];
        const bool valid = nl >= 0;                                 // valid_neighbor, :77
        neq += (valid && nl != me) ? 1 : 0;                         // :80-81
        all_eq = all_eq && (nl == me || !valid);                    // :92-93
    }
    if (bound) bound[i] = neq > 0;
    if (plain) plain[i] = all_eq;
    if (cnt) cnt[i] = neq;
}

// ---- boundary-IoU evaluation: masks + masked intersection / output / target histograms in one pass ---------------------------
// (tool/test.py:392-417 with get_boundary_mask basic_operators.py:69-97 and intersectionAndUnion util/common_util.py:25-37)
// hist[mask][what][class], mask 0 = boundary points, 1 = plain points; what 0 = intersection, 1 = output area, 2 = target area
__global__ __launch_bounds__(256) void boundary_iou_kernel(int n, int k, int ncls, long long ignore, const long long* __restrict__ pred,
                                                           const long long* __restrict__ labels, const int* __restrict__ nidx,
                                                           unsigned long long* __restrict__ hist)
{
    extern __shared__ unsigned lh[];                                // [2][3][ncls]
    for (int e = threadIdx.x; e < 6 * ncls; e += 256) lh[e] = 0u;
    __syncthreads();
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const long long me = labels[i];
        bool any_neq = false, all_eq = true;
        for (int j = 0; j < k; j++) {
            const long long nl = labels[nidx[(size_t)i * k + j]];
            const bool valid = nl >= 0;                             // :77
            any_neq = any_neq || (valid && nl != me);               // :80-81
            all_eq = all_eq && (nl == me || !valid);                // :92-93
        }
        const long long out = (me == ignore) ? ignore : pred[i];    // common_util.py:31
#pragma unroll
        for (int mk = 0; mk < 2; mk++) {
            if (mk == 0 ? any_neq : all_eq) {
                unsigned* h = lh + mk * 3 * ncls;
                if (out == me && out >= 0 && out < ncls) atomicAdd(h + out, 1u);                 // :32-33
                if (out >= 0 && out < ncls) atomicAdd(h + ncls + out, 1u);                       // :34
                if (me >= 0 && me < ncls) atomicAdd(h + 2 * ncls + me, 1u);                      // :35
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 6 * ncls; e += 256) if (lh[e]) atomicAdd(hist + e, (unsigned long long)lh[e]);
}

template <int G>
int launch_contrast(bool fwd, int m, int nsample, int d, const float* feat, const int* amax, const int* nidx, float inv_t, float weight,
                    int n_valid, int tf_variant, float kl_thr,
                    float* per_point, int* point_mask, const float* stats, const float* grad_loss, float* grad_feat, hipStream_t st)
{
    const dim3 grid(cbl_div_up((long long)m * G, 256)), block(256);
#define CBL_CONTRAST_DV(DV)                                                                                                                  \
    if (fwd) hipLaunchKernelGGL((contrast_fwd_kernel<G, DV>), grid, block, 0, st, m, nsample, feat, amax, nidx, inv_t, n_valid, tf_variant, kl_thr, per_point, point_mask); \
    else     hipLaunchKernelGGL((contrast_bwd_kernel<G, DV>), grid, block, 0, st, m, nsample, feat, amax, nidx, inv_t, weight, n_valid, tf_variant, kl_thr, stats, grad_loss, grad_feat, \
                                per_point, point_mask)
    switch (d) {
        case 4:  CBL_CONTRAST_DV(1); break;
        case 8:  CBL_CONTRAST_DV(2); break;
        case 16: CBL_CONTRAST_DV(4); break;
        case 32: CBL_CONTRAST_DV(8); break;
        case 64: CBL_CONTRAST_DV(16); break;
        default: return CBL_ERR_UNSUPPORTED;
    }
#undef CBL_CONTRAST_DV
    return cbl_status();
}

int dispatch_contrast(bool fwd, int m, int nsample, int d, const float* feat, const int* amax, const int* nidx, float temperature, float weight,
                      int n_valid, int tf_variant,
                      float* per_point, int* point_mask, const float* stats, const float* grad_loss, float* grad_feat, hipStream_t st, float kl_thr = 0.f)
{
    const int ns = nsample - 1;
    const float inv_t = 1.0f / temperature;
    if (ns <= 16) return launch_contrast<16>(fwd, m, nsample, d, feat, amax, nidx, inv_t, weight, n_valid, tf_variant, kl_thr, per_point, point_mask, stats, grad_loss, grad_feat, st);
    if (ns <= 32) return launch_contrast<32>(fwd, m, nsample, d, feat, amax, nidx, inv_t, weight, n_valid, tf_variant, kl_thr, per_point, point_mask, stats, grad_loss, grad_feat, st);
    return launch_contrast<64>(fwd, m, nsample, d, feat, amax, nidx, inv_t, weight, n_valid, tf_variant, kl_thr, per_point, point_mask, stats, grad_loss, grad_feat, st);
}

}  // namespace

// shared with cbl_pairs.hip
int cbl_contrast_finalize_launch(int m, float weight, const float* per_point, const int* point_mask, float* stats, float* loss, hipStream_t st)
{
    hipLaunchKernelGGL(contrast_finalize_kernel, dim3(1), dim3(1024), 0, st, m, weight, per_point, point_mask, stats, loss);
    return cbl_status();
}

CBL_EXPORT int cbl_subscene_label(int m, int kr, int num_classes, const long long* target, const int* neighbor_idx, float* out, void* stream)
{
    if (m < 0 || kr <= 0 || num_classes <= 0 || num_classes > 64) return CBL_ERR_BAD_ARG;
    if (m == 0) return CBL_OK;
    if (!target || !neighbor_idx || !out) return CBL_ERR_BAD_ARG;
    hipLaunchKernelGGL(subscene_label_kernel, dim3(cbl_div_up((long long)m * 64, 256)), dim3(256), 0, cbl_stream(stream), m, kr, num_classes, target, neighbor_idx, 0x7fffffff, 0, out);
    return cbl_status();
}

CBL_EXPORT int cbl_tf_scene_label(int m, int n_valid, int k, int num_classes, const long long* point_labels, const int* scene_neighbor, int by_valid,
                                  float* out, void* stream)
{
    if (m < 0 || n_valid < 0 || k <= 0 || num_classes <= 0 || num_classes > 64) return CBL_ERR_BAD_ARG;
    if (m == 0) return CBL_OK;
    if (!point_labels || !scene_neighbor || !out) return CBL_ERR_BAD_ARG;
    hipLaunchKernelGGL(subscene_label_kernel, dim3(cbl_div_up((long long)m * 64, 256)), dim3(256), 0, cbl_stream(stream), m, k, num_classes, point_labels,
                       scene_neighbor, n_valid, by_valid, out);
    return cbl_status();
}

CBL_EXPORT int cbl_label_argmax(int m, int num_classes, const float* labels, int* amax, void* stream)
{
    if (m < 0 || num_classes <= 0) return CBL_ERR_BAD_ARG;
    if (m == 0) return CBL_OK;
    if (!labels || !amax) return CBL_ERR_BAD_ARG;
    hipLaunchKernelGGL(label_argmax_kernel, dim3(cbl_div_up(m, 256)), dim3(256), 0, cbl_stream(stream), m, num_classes, labels, amax);
    return cbl_status();
}

static int point_contrast_forward_impl(int m, int nsample, int d, const float* features, const int* amax, int flags, const int* neighbor_idx,
                                       float temperature, float weight, float* per_point, int* point_mask, float* stats, float* loss, void* stream)
{
    if (m <= 0 || nsample < 2 || nsample > 65 || d <= 0 || !(temperature > 0.f)) return CBL_ERR_BAD_ARG;
    if (!features || !amax || !neighbor_idx || !per_point || !point_mask || !stats || !loss) return CBL_ERR_BAD_ARG;
    if (!cbl_host_aligned16(features)) return CBL_ERR_BAD_ARG;
    hipStream_t st = cbl_stream(stream);
    const int rc = dispatch_contrast(true, m, nsample, d, features, amax, neighbor_idx, temperature, weight, 0x7fffffff, flags, per_point, point_mask, nullptr, nullptr, nullptr, st);
    if (rc) return rc;
    hipLaunchKernelGGL(contrast_finalize_kernel, dim3(1), dim3(1024), 0, st, m, weight, per_point, point_mask, stats, loss);
    return cbl_status();
}

CBL_EXPORT int cbl_point_contrast_forward(int m, int nsample, int d, const float* features, const int* amax, const int* neighbor_idx,
                                          float temperature, float weight, float* per_point, int* point_mask, float* stats, float* loss, void* stream)
{
    return point_contrast_forward_impl(m, nsample, d, features, amax, 0, neighbor_idx, temperature, weight, per_point, point_mask, stats, loss, stream);
}

CBL_EXPORT int cbl_point_contrast_forward_l64(int m, int nsample, int d, const float* features, const long long* labels, const int* neighbor_idx,
                                              float temperature, float weight, float* per_point, int* point_mask, float* stats, float* loss, void* stream)
{
    return point_contrast_forward_impl(m, nsample, d, features, reinterpret_cast<const int*>(labels), 2, neighbor_idx, temperature, weight, per_point, point_mask,
                                       stats, loss, stream);
}

CBL_EXPORT int cbl_tf_contrast_forward(int m, int n_valid, int nsample, int d, const float* features, const int* labels, const int* neighbors,
                                       float temperature, float weight, float* per_point, int* point_mask, float* stats, float* loss, void* stream)
{
    if (m <= 0 || n_valid < 0 || nsample < 2 || nsample > 65 || d <= 0 || !(temperature > 0.f)) return CBL_ERR_BAD_ARG;
    if (!features || !labels || !neighbors || !per_point || !point_mask || !stats || !loss) return CBL_ERR_BAD_ARG;
    if (!cbl_host_aligned16(features)) return CBL_ERR_BAD_ARG;
    hipStream_t st = cbl_stream(stream);
    const int rc = dispatch_contrast(true, m, nsample, d, features, labels, neighbors, temperature, weight, n_valid, 1, per_point, point_mask, nullptr, nullptr, nullptr, st);
    if (rc) return rc;
    hipLaunchKernelGGL(contrast_finalize_kernel, dim3(1), dim3(1024), 0, st, m, weight, per_point, point_mask, stats, loss);
    return cbl_status();
}

CBL_EXPORT int cbl_tf_contrast_backward(int m, int n_valid, int nsample, int d, const float* features, const int* labels, const int* neighbors,
                                        float temperature, float weight, const float* stats, const float* grad_loss, float* grad_features, void* stream)
{
    if (m <= 0 || n_valid < 0 || nsample < 2 || nsample > 65 || d <= 0 || !(temperature > 0.f)) return CBL_ERR_BAD_ARG;
    if (!features || !labels || !neighbors || !stats || !grad_loss || !grad_features) return CBL_ERR_BAD_ARG;
    if (!cbl_host_aligned16(features)) return CBL_ERR_BAD_ARG;
    return dispatch_contrast(false, m, nsample, d, features, labels, neighbors, temperature, weight, n_valid, 1, nullptr, nullptr, stats, grad_loss, grad_features, cbl_stream(stream));
}

CBL_EXPORT int cbl_point_contrast_backward(int m, int nsample, int d, const float* features, const int* amax, const int* neighbor_idx,
                                           float temperature, float weight, const float* stats, const float* grad_loss, float* grad_features, void* stream)
{
    if (m <= 0 || nsample < 2 || nsample > 65 || d <= 0 || !(temperature > 0.f)) return CBL_ERR_BAD_ARG;
    if (!features || !amax || !neighbor_idx || !stats || !grad_loss || !grad_features) return CBL_ERR_BAD_ARG;
    if (!cbl_host_aligned16(features)) return CBL_ERR_BAD_ARG;
    return dispatch_contrast(false, m, nsample, d, features, amax, neighbor_idx, temperature, weight, 0x7fffffff, 0, nullptr, nullptr, stats, grad_loss, grad_features, cbl_stream(stream));
}

namespace {
// grad_features = grad_unit * (grad_loss * weight / count); all zeros when no point qualified (the loss was the constant 0)
__global__ __launch_bounds__(256) void contrast_grad_scale_kernel(long long total, const float* __restrict__ unit, const float* __restrict__ stats,
                                                                  const float* __restrict__ grad_loss, float weight, float* __restrict__ out)
{
    const float count = stats[1];
    const float sc = count > 0.f ? grad_loss[0] * weight / count : 0.f;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) out[e] = count > 0.f ? unit[e] * sc : 0.f;
}
}  // namespace

static int contrast_forward_grad(int m, int n_valid, int tf_variant, int nsample, int d, const float* features, const int* labels, const int* neighbors,
                                 float temperature, float weight, float* per_point, int* point_mask, float* stats, float* loss, float* grad_unit, void* stream,
                                 float kl_thr = 0.f)
{
    if (m <= 0 || n_valid < 0 || nsample < 2 || nsample > 65 || d <= 0 || !(temperature > 0.f)) return CBL_ERR_BAD_ARG;
    if (!features || !labels || !neighbors || !per_point || !point_mask || !stats || !loss || !grad_unit) return CBL_ERR_BAD_ARG;
    if (!cbl_host_aligned16(features)) return CBL_ERR_BAD_ARG;
    hipStream_t st = cbl_stream(stream);
    const int rc = dispatch_contrast(false, m, nsample, d, features, labels, neighbors, temperature, weight, n_valid, tf_variant, per_point, point_mask,
                                     nullptr, nullptr, grad_unit, st, kl_thr);
    if (rc) return rc;
    hipLaunchKernelGGL(contrast_finalize_kernel, dim3(1), dim3(1024), 0, st, m, weight, per_point, point_mask, stats, loss);
    return cbl_status();
}

CBL_EXPORT int cbl_point_contrast_forward_grad(int m, int nsample, int d, const float* features, const int* amax, const int* neighbor_idx,
                                               float temperature, float weight, float* per_point, int* point_mask, float* stats, float* loss,
                                               float* grad_unit, void* stream)
{
    return contrast_forward_grad(m, 0x7fffffff, 0, nsample, d, features, amax, neighbor_idx, temperature, weight, per_point, point_mask, stats, loss, grad_unit, stream);
}

CBL_EXPORT int cbl_point_contrast_forward_grad_l64(int m, int nsample, int d, const float* features, const long long* labels, const int* neighbor_idx,
                                                   float temperature, float weight, float* per_point, int* point_mask, float* stats, float* loss,
                                                   float* grad_unit, void* stream)
{
    return contrast_forward_grad(m, 0x7fffffff, 2, nsample, d, features, reinterpret_cast<const int*>(labels), neighbor_idx, temperature, weight, per_point, point_mask,
                                 stats, loss, grad_unit, stream);
}

CBL_EXPORT int cbl_tf_contrast_forward_grad(int m, int n_valid, int nsample, int d, const float* features, const int* labels, const int* neighbors,
                                            float temperature, float weight, float* per_point, int* point_mask, float* stats, float* loss,
                                            float* grad_unit, void* stream)
{
    return contrast_forward_grad(m, n_valid, 1, nsample, d, features, labels, neighbors, temperature, weight, per_point, point_mask, stats, loss, grad_unit, stream);
}

// sample 'labelkl<thr>' of the TF head (config/s3dis.py:162-163, README row "ConvNet + CBL (kl)"): positives by the KL divergence of SOFT labels
CBL_EXPORT int cbl_tf_contrast_forward_kl(int m, int n_valid, int nsample, int d, const float* features, const float* soft_labels, int num_classes,
                                          float kl_threshold, const int* neighbors, float temperature, float weight, float* per_point, int* point_mask,
                                          float* stats, float* loss, void* stream)
{
    if (m <= 0 || n_valid < 0 || nsample < 2 || nsample > 65 || d <= 0 || !(temperature > 0.f) || num_classes <= 0 || num_classes > 255) return CBL_ERR_BAD_ARG;
    if (!features || !soft_labels || !neighbors || !per_point || !point_mask || !stats || !loss) return CBL_ERR_BAD_ARG;
    if (!cbl_host_aligned16(features)) return CBL_ERR_BAD_ARG;
    hipStream_t st = cbl_stream(stream);
    const int rc = dispatch_contrast(true, m, nsample, d, features, reinterpret_cast<const int*>(soft_labels), neighbors, temperature, weight, n_valid,
                                     1 | (num_classes << 8), per_point, point_mask, nullptr, nullptr, nullptr, st, kl_threshold);
    if (rc) return rc;
    hipLaunchKernelGGL(contrast_finalize_kernel, dim3(1), dim3(1024), 0, st, m, weight, per_point, point_mask, stats, loss);
    return cbl_status();
}

CBL_EXPORT int cbl_tf_contrast_forward_grad_kl(int m, int n_valid, int nsample, int d, const float* features, const float* soft_labels, int num_classes,
                                               float kl_threshold, const int* neighbors, float temperature, float weight, float* per_point, int* point_mask,
                                               float* stats, float* loss, float* grad_unit, void* stream)
{
    if (num_classes <= 0 || num_classes > 255 || !soft_labels) return CBL_ERR_BAD_ARG;
    return contrast_forward_grad(m, n_valid, 1 | (num_classes << 8), nsample, d, features, reinterpret_cast<const int*>(soft_labels), neighbors, temperature, weight,
                                 per_point, point_mask, stats, loss, grad_unit, stream, kl_threshold);
}

CBL_EXPORT int cbl_contrast_grad_scale(long long total, const float* grad_unit, const float* stats, const float* grad_loss, float weight,
                                       float* grad_features, void* stream)
{
    if (total < 0) return CBL_ERR_BAD_ARG;
    if (total == 0) return CBL_OK;
    if (!grad_unit || !stats || !grad_loss || !grad_features) return CBL_ERR_BAD_ARG;
    hipLaunchKernelGGL(contrast_grad_scale_kernel, dim3(cbl_grid_for(total, 256, 2048)), dim3(256), 0, cbl_stream(stream), total, grad_unit, stats, grad_loss, weight, grad_features);
    return cbl_status();
}

CBL_EXPORT int cbl_boundary_mask(int n, int k, const long long* labels, const int* neighbor_idx, unsigned char* bound, unsigned char* plain, int* cnt, void* stream)
{
    if (n < 0 || k <= 0) return CBL_ERR_BAD_ARG;
    if (n == 0) return CBL_OK;
    if (!labels || !neighbor_idx) return CBL_ERR_BAD_ARG;
    hipLaunchKernelGGL(boundary_mask_kernel, dim3(cbl_div_up(n, 256)), dim3(256), 0, cbl_stream(stream), n, k, labels, neighbor_idx, bound, plain, cnt);
    return cbl_status();
}

CBL_EXPORT int cbl_boundary_iou(int n, int k, int num_classes, long long ignore_label, const long long* pred, const long long* labels,
                                const int* neighbor_idx, unsigned long long* hist, void* stream)
{
    if (n < 0 || k <= 0 || num_classes <= 0 || num_classes > 2048) return CBL_ERR_BAD_ARG;
    if (!hist) return CBL_ERR_BAD_ARG;
    if (n == 0) return CBL_OK;
    if (!pred || !labels || !neighbor_idx) return CBL_ERR_BAD_ARG;
    hipLaunchKernelGGL(boundary_iou_kernel, dim3(cbl_grid_for(n, 256, 1024)), dim3(256), sizeof(unsigned) * 6 * (size_t)num_classes, cbl_stream(stream),
                       n, k, num_classes, ignore_label, pred, labels, neighbor_idx, hist);
    return cbl_status();
}
