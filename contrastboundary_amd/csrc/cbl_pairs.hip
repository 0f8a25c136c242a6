// CBL pair mining + soft-NN loss with an ATOMIC-FREE gradient (SURVEY.md §7 hard part 6).
//   ContrastHead.point_contrast     /root/reference/pytorch/model/heads.py:185-246  (posmask_cnt :145-149, dist_l2 :116-119, contrast_softnn :151-165)
//   TF contrast_head                /root/reference/tensorflow/models/heads/head.py:462-807 (flags bit 0; the differences are listed in cbl.hip)
//
// d loss / d f_t has two halves: the pairs in which t is the CENTRE, coef_tj (f_t - f_j) over t's own neighbour list, and the pairs in which
// t is a NEIGHBOUR, coef_it (f_t - f_i) over the points i that list t.  Round 1 scattered the second half with float atomics (16.9 M of
// them, executed on the memory side of the fabric: 68 MB written for a 5 MB gradient).  Here
//   pass A (contrast_pairs_kernel, one wave per point): mining, loss term, the scalar coef of every pair (m, nsample) and the centre half
//          of the gradient, reduced in registers;
//   pass B (contrast_gather_kernel, one wave per point): the neighbour half as a GATHER over the transposed neighbour table
//          (cbl_neighbor_transpose), plus the centre half, times the global factor — plain stores only, no zero fill, deterministic.
// Lane layout of both passes: a feature row of d floats is read by LR = d / 4 consecutive lanes (16 B each: one coalesced 4*d-byte
// request per row instead of one 64 B sector per lane and float4), PP = 64 / LR rows per load instruction.
#include "cbl_common.h"
#include "wave_ops.h"

namespace {


// sum over the LR lanes that share a row (LR consecutive lanes, LR | 16): every one of them ends up with the total
template <int LR> __device__ __forceinline__ float row_lanes_sum(float v)
{
    if (LR >= 2) v = dpp_add0<0xB1>(v);                      // quad_perm [1,0,3,2]
    if (LR >= 4) v = dpp_add0<0x4E>(v);                      // quad_perm [2,3,0,1]
    if (LR >= 8) v = dpp_add0<0x141>(v);                     // row_half_mirror
    if (LR >= 16) v = dpp_add0<0x140>(v);                    // row_mirror
    return v;
}
// sum over the PP = 64 / LR row slots (lanes with the same position inside their row)
template <int LR> __device__ __forceinline__ float slots_sum(float v)
{
    if (LR <= 8) v = dpp_add0<0x128>(v);                     // row_ror:8  (lane ^ 8 inside a row of 16)
    if (LR <= 4) v = dpp_add0<0x124>(v);                     // row_ror:4  (after ^8 every lane l holds l and l^8: rotate by 4 adds l^4, l^12)
    if (LR <= 2) v = dpp_add0<0x122>(v);
    if (LR <= 1) v = dpp_add0<0x121>(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// flags: bit 0 TF flavour, bit 1 labels are int64 (read through their low words), bit 2 contrast 'nce' instead of 'softnn'
// (heads.py:167-183: one -log(e_j / (e_j + sum of negatives)) per POSITIVE, averaged over all positives — point_mask then holds the
// point's number of positives; TF head.py:773-795 without 'S' / masking: -sum over positives of log(e_j / sum of valid + eps) per point),
// bits 8..15 ncls > 0: soft labels + KL positives
// v_sqrt_f32 / v_exp_f32 / v_rcp_f32 (1 ulp) instead of the correctly rounded library sequences (8-20 vector instructions each, per pair):
// ~1e-7 relative on every term, the contract of the loss and its gradient is 1e-4
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

template <int LR, int UM, bool GRAD>
__global__ __launch_bounds__(256) void contrast_pairs_kernel(unsigned m, int nsample, const float4* __restrict__ feat, const int* __restrict__ amax,
                                                             const int* __restrict__ nidx, const int* __restrict__ order, float inv_temperature, int n_valid,
                                                             int flags, float kl_thr, const unsigned char* __restrict__ roles,
                                                             const unsigned char* __restrict__ sample_valid, float* __restrict__ per_point,
                                                             int* __restrict__ point_mask, float* __restrict__ coef, float4* __restrict__ grad_own)
{
    constexpr int PP = 64 / LR;
    const int tf_variant = flags & 1, ls = 1 + ((flags >> 1) & 1), nce = (flags >> 2) & 1, sep = (flags >> 3) & 1, ncls = (flags >> 8) & 0xff;
    const int ns = nsample - 1;                                     // self column dropped, heads.py:195-196 / head.py:560
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform values in scalar registers: scalar loads, uniform branches
    const unsigned nwg = (m + 3) >> 2;                              // 4 points per workgroup
    // A point is three dependent round trips (sequence slot -> point id -> its neighbour ids -> their labels and rows) in front of ~300 clocks of
    // arithmetic; the first two are prefetched: the point id two trips ahead, its neighbour ids one trip ahead (values only: nothing branches on
    // them before their own trip).  Measured and not kept: labels (one trip ahead) deciding whether a point's rows are requested at all (39 us
    // against 35), labels and rows one trip ahead (38 us): both cost the fifth wave per SIMD, and the waves cover each other's round trips
    // better than a wave covers its own.
    const unsigned vend = 8 * cbl_xcd_per(nwg), vstep = gridDim.x;
    const bool colr = lane < ns;
    auto point_of = [&](unsigned v) -> int {
        const unsigned r = (v < vend ? cbl_xcd_slot(v, nwg) : 0u) * 4 + wave;
        const bool ok = v < vend && r < m;
        const int pt = order ? order[ok ? r : 0u] : (int)r;
        return __builtin_amdgcn_readfirstlane(ok ? pt : -1);
    };
    auto ids_of = [&](int pt) -> int { return nidx[(size_t)(pt < 0 ? 0 : pt) * nsample + 1 + (colr ? lane : 0)]; };
    int iB = point_of(blockIdx.x);
    int rawB = ids_of(iB);
    int iA = point_of(blockIdx.x + vstep);
    for (unsigned v = blockIdx.x; v < vend; v += vstep) {
        const int i = __builtin_amdgcn_readfirstlane(iB), raw = rawB;
        iB = iA; rawB = ids_of(iB); iA = point_of(v + 2 * vstep);
        if (i < 0) continue;
        // ---- mining, lane = neighbour column
        const bool real = raw >= 0 && raw < n_valid;
        const int nbr_row = real ? raw : 0;
        // the neighbours' rows are requested HERE, together with their labels (both need nothing but the ids): whether the point has a loss at all is
        // only known a round trip later, and 63 % of the S-room scene's points have none — their rows are fetched for nothing, but a point that has
        // one no longer pays labels and rows one after the other
        const int s = lane / LR, q = lane % LR;
        float4 fjv[UM];
#pragma unroll
        for (int u = 0; u < UM; u++) {
            const int j = u * PP + s;
            fjv[u] = feat[(size_t)__shfl(nbr_row, j < ns ? j : 0) * LR + q];
        }
        const float4 fi = feat[(size_t)i * LR + q];
        bool nb_r, pos_r;
        if (ncls) {                                                  // collect_labels head.py:498-511, calc_dist 'kl' :189-191
            const float* __restrict__ soft = reinterpret_cast<const float*>(amax);
            float kl = 0.f;
            for (int c = 0; c < ncls; c++) {
                const float pi = soft[(size_t)i * ncls + c], pj = real ? soft[(size_t)nbr_row * ncls + c] : 0.f;
                if (pi > 0.f) kl += pi * logf(pi / fmaxf(pj, 1e-12f));
            }
            nb_r = colr && real;
            pos_r = nb_r && (kl < kl_thr);
        } else {
            const int my = amax[(size_t)i * ls], nl = amax[(size_t)nbr_row * ls];
            nb_r = colr && real && (!tf_variant || (my >= 0 && nl >= 0));
            pos_r = nb_r && (nl == my);                              // posmask_cnt :145-149 / head.py:538
        }
        if (roles) {                                                 // sample_labels head.py:560-625: columns that are not label-mined (wave-uniform branch)
            const int role = colr ? roles[lane] : 0;
            if (role) {                                              // 'nn<k>': positive, 'rand<n>': negative, both valid whatever they point at (:603-617);
                nb_r = colr && (role != CBL_ROLE_NEG_REJECT || !sample_valid || sample_valid[(size_t)i * ns + lane] != 0);   // 'rand<n>R': unless it is one of the neighbours
                pos_r = nb_r && role == CBL_ROLE_POS;
            }
        }
        const unsigned long long nbmask = __ballot(nb_r), posmask = __ballot(pos_r), realmask = __ballot(real);
        const int cnt = __popcll(posmask), nvalid = __popcll(nbmask);
        const bool valid = cnt > 0 && cnt < nvalid;                  // :212-213 / solve_samples_mask head.py:621-640 (wave-uniform)
        if (!valid) {                                                // neither loss nor gradient: the labels decide before any feature is read
            if (lane == 0) { per_point[i] = 0.f; point_mask[i] = 0; }
            if (GRAD) {
                if (lane < nsample) coef[(size_t)i * nsample + lane] = 0.f;
                if (nsample > 64 && lane == 0) coef[(size_t)i * nsample + 64] = 0.f;
                if (lane < LR) grad_own[(size_t)i * LR + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            continue;
        }
        // ---- distances, lane = (row slot s, part q of the row)
        float shadow_d = 0.f;
        if (tf_variant) shadow_d = sqrtf(fmaxf(row_lanes_sum<LR>((fi.x * fi.x + fi.y * fi.y) + (fi.z * fi.z + fi.w * fi.w)), 1e-12f));
        float4 diff[UM]; float dist[UM], ex[UM];
        bool isnb[UM], ispos[UM];
        float mxl = -INFINITY;
#pragma unroll
        for (int u = 0; u < UM; u++) {
            const int j = u * PP + s;
            const bool col = j < ns;
            const int jj = col ? j : 0;
            const float4 fj = fjv[u];
            diff[u] = make_float4(fi.x - fj.x, fi.y - fj.y, fi.z - fj.z, fi.w - fj.w);
            const float acc = row_lanes_sum<LR>((diff[u].x * diff[u].x + diff[u].y * diff[u].y) + (diff[u].z * diff[u].z + diff[u].w * diff[u].w));
            dist[u] = fast_sqrt(tf_variant ? fmaxf(acc, 1e-12f) : acc + 1e-12f);     // head.py:184-185 / dist_l2 heads.py:116-119
            isnb[u] = col && ((nbmask >> jj) & 1ull);
            ispos[u] = col && ((posmask >> jj) & 1ull);
            // shadow columns of the TF flavour gather a zero feature row (tf_gather, basic_operators.py:381-410): they DO enter the max-shift
            // (head.py:752), and an 'nn<k>' column keeps them as a positive
            if (tf_variant && !((realmask >> jj) & 1ull)) { dist[u] = shadow_d; diff[u] = fi; }
            ex[u] = (isnb[u] || (tf_variant && col)) ? -dist[u] : -INFINITY;
            mxl = fmaxf(mxl, ex[u]);
        }
        const float mx = group_max<64>(mxl);                         // :153
        float pl = 0.f, al = 0.f, nl = 0.f;
#pragma unroll
        for (int u = 0; u < UM; u++) {
            ex[u] = isnb[u] ? fast_exp((ex[u] - mx) * inv_temperature) : 0.f;       // shift, then / T (:153-155)
            pl += ispos[u] ? ex[u] : 0.f; al += ex[u]; nl += ispos[u] ? 0.f : ex[u];
        }
        const float P = group_sum<64>(pl) * (1.0f / LR), A = group_sum<64>(al) * (1.0f / LR);   // every pair is held by LR lanes
        // the negatives' sum where a variant uses it by itself ('nce', margin 'S'): summed as the reference sums it (heads.py:171, head.py:757) — A - P loses
        // it to cancellation once the negatives are 1e-7 of the positives
        const float Nsum = (nce || sep) ? group_sum<64>(nl) * (1.0f / LR) : 0.f;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!nce) {
            // margin 'S' (head.py:759-760): pos / max(neg, eps) instead of pos / (pos + neg)
            const float Nn = sep ? Nsum : A - P, Nc = fmaxf(Nn, 1e-12f);
            const float ratio = sep ? P / Nc : P / A;
            if (lane == 0) { per_point[i] = -logf(ratio + 1e-12f); point_mask[i] = 1; }          // contrast_softnn :161-163
            if (!GRAD) continue;
            // ---- gradient coefficients: d term / d dist_j, then / dist_j for the direction (f_i - f_j) / dist_j
            // d(P / A) / d e_j = ((j positive ? A : 0) - P) / A^2, written as two quotients by A: A^2 underflows in fp32 once A < 1e-19 — every valid neighbour
            // more than 44 T farther away than a masked column that holds the maximum of the shift (head.py:752) — and the coefficient became inf / NaN
            // where the reference's autodiff (x / y / y) stays finite
            const float invA = 1.0f / A;
            const float base = inv_temperature / (ratio + 1e-12f), es = sep ? 1.0f : invA;
            const float xpos = sep ? 1.0f / Nc : (A - P) * invA, xneg = sep ? (Nn > 1e-12f ? -P / (Nc * Nc) : 0.f) : -P * invA;
            if (lane == 0) coef[(size_t)i * nsample] = 0.f;          // the self column takes no part
#pragma unroll
            for (int u = 0; u < UM; u++) {
                const int j = u * PP + s;
                float c = isnb[u] ? (ex[u] * es) * (ispos[u] ? xpos : xneg) * base * fast_rcp(dist[u]) : 0.f;
                if (tf_variant && dist[u] <= 1e-6f) c = 0.f;        // sqrt(max(s, 1e-12)): flat below the clamp
                if (q == 0 && j < ns) coef[(size_t)i * nsample + 1 + j] = c;
                g.x += c * diff[u].x; g.y += c * diff[u].y; g.z += c * diff[u].z; g.w += c * diff[u].w;
            }
        } else {
            // ---- contrast 'nce'.  pytorch (heads.py:167-183): term_j = -log(e_j / (e_j + N)), N = sum of the negatives' e, one term per
            // positive; TF (head.py:773-795): -sum over positives of log(e_j / A + eps) per point
            const float N = Nsum;
            float tl = 0.f, ql = 0.f;
#pragma unroll
            for (int u = 0; u < UM; u++) {
                if (ispos[u]) {
                    if (tf_variant && sep) {                        // 'S': under_j = e_j + N (head.py:783-785), eps inside the log (:793)
                        const float un = ex[u] + N, r = ex[u] / un;
                        tl += -logf(r + 1e-12f); ql += r / ((r + 1e-12f) * un);              // e_j / ((r + eps) un^2) without the square (underflow, as above)
                    }
                    else if (tf_variant) { const float r = ex[u] / A; tl += -logf(r + 1e-12f); ql += r / (r + 1e-12f); }
                    else                 { tl += -logf(ex[u] / (ex[u] + N)); ql += 1.0f / (ex[u] + N); }
                }
            }
            const float term = group_sum<64>(tl) * (1.0f / LR), Q = group_sum<64>(ql) * (1.0f / LR);
            if (lane == 0) { per_point[i] = term; point_mask[i] = tf_variant ? 1 : cnt; }        // finalize: mean over points (TF) / over positives (pytorch)
            if (!GRAD) continue;
            if (lane == 0) coef[(size_t)i * nsample] = 0.f;
#pragma unroll
            for (int u = 0; u < UM; u++) {
                const int j = u * PP + s;
                float c = 0.f;
                if (isnb[u]) {
                    if (tf_variant && sep) {
                        const float un = ex[u] + N, r = ex[u] / un;
                        c = (ispos[u] ? inv_temperature * r * (N / un) / (r + 1e-12f) : -inv_temperature * ex[u] * Q) / dist[u];
                    }
                    else if (tf_variant) { const float r = ex[u] / A; c = inv_temperature * ((ispos[u] ? r / (r + 1e-12f) : 0.f) - r * Q) / dist[u]; }
                    else                 c = (ispos[u] ? inv_temperature * N / (ex[u] + N) : -inv_temperature * ex[u] * Q) / dist[u];
                }
                if (tf_variant && dist[u] <= 1e-6f) c = 0.f;
                if (q == 0 && j < ns) coef[(size_t)i * nsample + 1 + j] = c;
                g.x += c * diff[u].x; g.y += c * diff[u].y; g.z += c * diff[u].z; g.w += c * diff[u].w;
            }
        }
        g.x = slots_sum<LR>(g.x); g.y = slots_sum<LR>(g.y); g.z = slots_sum<LR>(g.z); g.w = slots_sum<LR>(g.w);
        if (s == 0) grad_own[(size_t)i * LR + q] = g;
    }
}

// pass B: grad[t] = (grad_own[t] + sum over the pairs p = (i, col) that list t of coef[p] (f_t - f_i)) * grad_loss * weight / count
template <int LR>
__global__ __launch_bounds__(256) void contrast_gather_kernel(unsigned m, CblFastDiv dv, const float4* __restrict__ feat, const float* __restrict__ coef,
                                                              const float4* __restrict__ grad_own, const int* __restrict__ order,
                                                              const int* __restrict__ inv_start, const int* __restrict__ inv_src,
                                                              const float* __restrict__ stats, const float* __restrict__ grad_loss, float weight,
                                                              float4* __restrict__ grad)
{
    constexpr int PP = 64 / LR;
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = lane / LR, q = lane % LR;
    const float count = stats[1];
    const float scale = count > 0.f ? grad_loss[0] * weight / count : 0.f;      // torch.mean(loss) * w (:241-243); 0 when no point qualified (:233)
    const unsigned nwg = (m + 3) >> 2;
    for (unsigned v = blockIdx.x; v < 8 * cbl_xcd_per(nwg); v += gridDim.x) {
        const unsigned r = cbl_xcd_slot(v, nwg) * 4 + wave;
        if (r >= m) continue;
        const int t = order ? order[r] : (int)r;
        const int s0 = inv_start[r], s1 = inv_start[r + 1];
        const float4 ft = feat[(size_t)t * LR + q];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (count > 0.f) {
            // lane e takes entry e of the list (its pair and that pair's coefficient) for 64 entries at a time: two round trips for all of them.
            // Most coefficients are zero (only points with a mixed neighbourhood have a loss: 37 % of the S-room scene, 13 of a target's 35 pairs,
            // none at all for 45 % of the targets), so the entries that carry one are packed to the front (ballot + ds_permute) and only their
            // source rows are gathered, GB at a time with every address known (a third round trip in all).
            // (Tried: persistent waves with bounds / pair ids / coefficients fetched three, two and one target ahead — not faster, 28 vs 26 us
            // for the stage: eight waves per SIMD hide the chain as well.)
            constexpr int GB = 4 * PP > 64 ? 64 : 4 * PP;          // entries per group of row loads in flight together
            for (int eb = s0; eb < s1; eb += 64) {
                const int e = eb + lane;
                const int p = inv_src[e < s1 ? e : s0];
                const float c = e < s1 ? coef[p] : 0.f;
                const unsigned long long live = __ballot(c != 0.f);
                if (live == 0ull) continue;                          // (wave-uniform)
                const int nnz = __popcll(live);
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(live >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)live, 0u));
                const int dst = (c != 0.f ? rank : 63) << 2;        // push: a lane with a coefficient sends (c, source point) to lane `rank`
                const int cbits = __builtin_amdgcn_ds_permute(dst, c != 0.f ? __float_as_int(c) : 0);
                const int sbits = __builtin_amdgcn_ds_permute(dst, c != 0.f ? (int)cbl_fastdiv((unsigned)p, dv) : 0);
                const float cp = lane < nnz ? __int_as_float(cbits) : 0.f;      // lanes >= nnz received nothing that counts
                const unsigned sp = lane < nnz ? (unsigned)sbits : 0u;
                for (int h0 = 0; h0 < nnz; h0 += GB) {               // (trip count wave-uniform)
#pragma unroll
                    for (int b0 = 0; b0 < GB; b0 += PP) {
                        const int src = (h0 + b0 + s) & 63;
                        const float ce = (h0 + b0 + s) < 64 ? __shfl(cp, src) : 0.f;
                        const unsigned ie = (unsigned)__shfl((int)sp, src);
                        // unconditional (a row fetched for a zero coefficient adds nothing): a load inside an `if` waits inside it, one round
                        // trip per step instead of one for all
                        const float4 fi = feat[(size_t)ie * LR + q];
                        acc.x += ce * (ft.x - fi.x); acc.y += ce * (ft.y - fi.y); acc.z += ce * (ft.z - fi.z); acc.w += ce * (ft.w - fi.w);
                    }
                }
            }
        }
        acc.x = slots_sum<LR>(acc.x); acc.y = slots_sum<LR>(acc.y); acc.z = slots_sum<LR>(acc.z); acc.w = slots_sum<LR>(acc.w);
        if (s == 0) {
            const float4 own = grad_own[(size_t)t * LR + q];
            grad[(size_t)t * LR + q] = make_float4((own.x + acc.x) * scale, (own.y + acc.y) * scale, (own.z + acc.z) * scale, (own.w + acc.w) * scale);
        }
    }
}

template <int LR, int UM>
int launch_pairs(unsigned g, hipStream_t st, int m, int nsample, const float* feat, const int* amax, const int* nidx, const int* order, float inv_t, int n_valid,
                 int flags, float kl_thr, const unsigned char* roles, const unsigned char* sample_valid, float* per_point, int* point_mask, float* coef, float* grad_own)
{
    g = coef ? cbl_persistent_grid(g, &contrast_pairs_kernel<LR, UM, true>, 256) : cbl_persistent_grid(g, &contrast_pairs_kernel<LR, UM, false>, 256);
    if (coef) hipLaunchKernelGGL((contrast_pairs_kernel<LR, UM, true>), dim3(g), dim3(256), 0, st, (unsigned)m, nsample, reinterpret_cast<const float4*>(feat), amax, nidx,
                                 order, inv_t, n_valid, flags, kl_thr, roles, sample_valid, per_point, point_mask, coef, reinterpret_cast<float4*>(grad_own));
    else hipLaunchKernelGGL((contrast_pairs_kernel<LR, UM, false>), dim3(g), dim3(256), 0, st, (unsigned)m, nsample, reinterpret_cast<const float4*>(feat), amax, nidx,
                            order, inv_t, n_valid, flags, kl_thr, roles, sample_valid, per_point, point_mask, nullptr, nullptr);
    return cbl_status();
}

template <int LR>
int dispatch_pairs_um(int U, unsigned g, hipStream_t st, int m, int nsample, const float* feat, const int* amax, const int* nidx, const int* order, float inv_t,
                      int n_valid, int flags, float kl_thr, const unsigned char* roles, const unsigned char* sample_valid, float* per_point, int* point_mask,
                      float* coef, float* grad_own)
{
#define CBL_PAIRS_UM(UM_) return launch_pairs<LR, UM_>(g, st, m, nsample, feat, amax, nidx, order, inv_t, n_valid, flags, kl_thr, roles, sample_valid, per_point, point_mask, coef, grad_own)
    if (U <= 1) CBL_PAIRS_UM(1);
    if (U <= 2) CBL_PAIRS_UM(2);
    if (U <= 3) CBL_PAIRS_UM(3);
    if (U <= 5) CBL_PAIRS_UM(5);
    if (U <= 8) CBL_PAIRS_UM(8);
    if (LR >= 16 && U <= 16) CBL_PAIRS_UM(16);
#undef CBL_PAIRS_UM
    return CBL_ERR_UNSUPPORTED;
}

// pass B without a transposed table (scenes beyond its 1 M-row limit): grad = grad_own * scale, then one float atomic per (pair with a coefficient,
// channel) — the reference's own index_select backward.  Not deterministic in the last bits; used only where the gather form is unavailable.
__global__ __launch_bounds__(256) void contrast_own_scale_kernel(long long total4, const float4* __restrict__ grad_own, const float* __restrict__ stats,
                                                                 const float* __restrict__ grad_loss, float weight, float4* __restrict__ grad)
{
    const float count = stats[1];
    const float scale = count > 0.f ? grad_loss[0] * weight / count : 0.f;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long long)gridDim.x * 256) {
        const float4 o = grad_own[e];
        grad[e] = make_float4(o.x * scale, o.y * scale, o.z * scale, o.w * scale);
    }
}
__global__ __launch_bounds__(256) void contrast_scatter_kernel(long long pairs, int lr, CblFastDiv dv, int n_valid, const float4* __restrict__ feat,
                                                               const float* __restrict__ coef, const int* __restrict__ nidx, const float* __restrict__ stats,
                                                               const float* __restrict__ grad_loss, float weight, float* __restrict__ grad)
{
    const float count = stats[1];
    if (!(count > 0.f)) return;
    const float scale = grad_loss[0] * weight / count;
    const long long total = pairs * lr;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long p = e / lr; const int q = (int)(e - p * lr);
        const float c = coef[p];
        if (c == 0.f) continue;
        const int t = nidx[p];
        if (t < 0 || t >= n_valid) continue;
        const long long i = (long long)cbl_fastdiv((unsigned)p, dv);
        const float4 ft = feat[(size_t)t * lr + q], fi = feat[(size_t)i * lr + q];
        float* g = grad + ((size_t)t * lr + q) * 4;
        unsafeAtomicAdd(g, scale * (c * (ft.x - fi.x))); unsafeAtomicAdd(g + 1, scale * (c * (ft.y - fi.y)));
        unsafeAtomicAdd(g + 2, scale * (c * (ft.z - fi.z))); unsafeAtomicAdd(g + 3, scale * (c * (ft.w - fi.w)));
    }
}

}  // namespace

// deterministic reduction of the per-point terms (cbl.hip)
int cbl_contrast_finalize_launch(int m, float weight, const float* per_point, const int* point_mask, float* stats, float* loss, hipStream_t st);

// sample strings beyond 'label' (head.py:560-625): sample_idx (m, nsample) = the self column followed by the concatenated sample columns,
// roles (nsample - 1) u8 per column (0 label-mined, 1 'nn' positive, 2 'rand' negative, 3 'rand..R' negative unless sample_valid says otherwise),
// sample_valid (m, nsample - 1) u8 read for role-3 columns only (NULL if there is none)
CBL_EXPORT int cbl_contrast_pairs_forward_samples(int m, int n_valid, int flags, int nsample, int d, const float* features, const void* labels, int num_classes,
                                                  float kl_threshold, const int* sample_idx, const unsigned char* roles, const unsigned char* sample_valid,
                                                  const int* order, float temperature, float weight, float* per_point, int* point_mask, float* stats,
                                                  float* loss, float* coef, float* grad_own, void* stream)
{
    if (m <= 0 || n_valid < 0 || nsample < 2 || nsample > 65 || d <= 0 || !(temperature > 0.f)) return CBL_ERR_BAD_ARG;
    if (!features || !labels || !sample_idx || !per_point || !point_mask || !stats || !loss) return CBL_ERR_BAD_ARG;
    if ((coef == nullptr) != (grad_own == nullptr)) return CBL_ERR_BAD_ARG;
    if ((flags & ~15) || num_classes < 0 || num_classes > 255) return CBL_ERR_BAD_ARG;
    if ((roles || sample_valid) && !(flags & 1)) return CBL_ERR_BAD_ARG;              // sample roles belong to the TF head
    const int* neighbor_idx = sample_idx;
    if (!cbl_host_aligned16(features) || (grad_own && !cbl_host_aligned16(grad_own))) return CBL_ERR_BAD_ARG;
    if (d % 4 || d > 64 || (d & (d - 1))) return CBL_ERR_UNSUPPORTED;
    hipStream_t st = cbl_stream(stream);
    const int fl = flags | (num_classes << 8);
    const float inv_t = 1.0f / temperature;
    const int* amax = reinterpret_cast<const int*>(labels);
    const int lr = d / 4, pp = 64 / lr, U = (nsample - 1 + pp - 1) / pp;
    unsigned g = cbl_round_up8(cbl_div_up(m, 4)); if (g > 256u * 32u) g = 256u * 32u;
    int rc;
#define CBL_PAIRS_LR(LR_) rc = dispatch_pairs_um<LR_>(U, g, st, m, nsample, features, amax, neighbor_idx, order, inv_t, n_valid, fl, kl_threshold, roles, sample_valid, per_point, point_mask, coef, grad_own)
    switch (lr) {
        case 1: CBL_PAIRS_LR(1); break;
        case 2: CBL_PAIRS_LR(2); break;
        case 4: CBL_PAIRS_LR(4); break;
        case 8: CBL_PAIRS_LR(8); break;
        case 16: CBL_PAIRS_LR(16); break;
        default: return CBL_ERR_UNSUPPORTED;
    }
#undef CBL_PAIRS_LR
    if (rc) return rc;
    return cbl_contrast_finalize_launch(m, weight, per_point, point_mask, stats, loss, st);
}

CBL_EXPORT int cbl_contrast_pairs_forward(int m, int n_valid, int flags, int nsample, int d, const float* features, const void* labels, int num_classes,
                                          float kl_threshold, const int* neighbor_idx, const int* order, float temperature, float weight,
                                          float* per_point, int* point_mask, float* stats, float* loss, float* coef, float* grad_own, void* stream)
{
    return cbl_contrast_pairs_forward_samples(m, n_valid, flags, nsample, d, features, labels, num_classes, kl_threshold, neighbor_idx, nullptr, nullptr, order,
                                              temperature, weight, per_point, point_mask, stats, loss, coef, grad_own, stream);
}

CBL_EXPORT int cbl_contrast_pairs_backward(int m, int nsample, int d, const float* features, const float* coef, const float* grad_own, const int* order,
                                           const int* inv_start, const int* inv_src, const float* stats, const float* grad_loss, float weight,
                                           float* grad_features, void* stream)
{
    if (m <= 0 || nsample < 2 || nsample > 65 || d <= 0) return CBL_ERR_BAD_ARG;
    if (!features || !coef || !grad_own || !inv_start || !inv_src || !stats || !grad_loss || !grad_features) return CBL_ERR_BAD_ARG;
    if (!cbl_host_aligned16(features) || !cbl_host_aligned16(grad_own) || !cbl_host_aligned16(grad_features)) return CBL_ERR_BAD_ARG;
    if (d % 4 || d > 64 || (d & (d - 1))) return CBL_ERR_UNSUPPORTED;
    hipStream_t st = cbl_stream(stream);
    unsigned g = cbl_round_up8(cbl_div_up(m, 4)); if (g > 256u * 32u) g = 256u * 32u;
    const CblFastDiv dv = cbl_fastdiv_make((unsigned)nsample);
#define CBL_GATHER_LR(LR_) hipLaunchKernelGGL((contrast_gather_kernel<LR_>), dim3(cbl_persistent_grid(g, &contrast_gather_kernel<LR_>, 256)), dim3(256), 0, st, (unsigned)m, dv, reinterpret_cast<const float4*>(features), coef, \
        reinterpret_cast<const float4*>(grad_own), order, inv_start, inv_src, stats, grad_loss, weight, reinterpret_cast<float4*>(grad_features))
    switch (d / 4) {
        case 1: CBL_GATHER_LR(1); break;
        case 2: CBL_GATHER_LR(2); break;
        case 4: CBL_GATHER_LR(4); break;
        case 8: CBL_GATHER_LR(8); break;
        case 16: CBL_GATHER_LR(16); break;
        default: return CBL_ERR_UNSUPPORTED;
    }
#undef CBL_GATHER_LR
    return cbl_status();
}

CBL_EXPORT int cbl_contrast_pairs_backward_atomic(int m, int n_valid, int nsample, int d, const float* features, const float* coef, const float* grad_own,
                                                  const int* neighbor_idx, const float* stats, const float* grad_loss, float weight,
                                                  float* grad_features, void* stream)
{
    if (m <= 0 || nsample < 2 || nsample > 65 || d <= 0) return CBL_ERR_BAD_ARG;
    if (!features || !coef || !grad_own || !neighbor_idx || !stats || !grad_loss || !grad_features) return CBL_ERR_BAD_ARG;
    if (!cbl_host_aligned16(features) || !cbl_host_aligned16(grad_own) || !cbl_host_aligned16(grad_features)) return CBL_ERR_BAD_ARG;
    if (d % 4 || d > 64 || (long long)m * nsample > 0x7fffffffll) return CBL_ERR_UNSUPPORTED;
    hipStream_t st = cbl_stream(stream);
    const long long total4 = (long long)m * (d / 4), pairs = (long long)m * nsample;
    hipLaunchKernelGGL(contrast_own_scale_kernel, dim3(cbl_grid_for(total4, 256)), dim3(256), 0, st, total4, reinterpret_cast<const float4*>(grad_own), stats, grad_loss,
                       weight, reinterpret_cast<float4*>(grad_features));
    hipLaunchKernelGGL(contrast_scatter_kernel, dim3(cbl_grid_for(pairs * (d / 4), 256)), dim3(256), 0, st, pairs, d / 4, cbl_fastdiv_make((unsigned)nsample),
                       n_valid < m ? n_valid : m, reinterpret_cast<const float4*>(features), coef, neighbor_idx, stats, grad_loss, weight, grad_features);
    return cbl_status();
}
