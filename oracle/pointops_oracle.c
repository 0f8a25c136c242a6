/*
 * pointops_oracle.c — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the algorithms of the reference's CUDA-only `pointops` kernels
 * (/root/reference/pytorch/lib/pointops/src/...).  It is the checker the HIP kernels are compared
 * against in tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; nothing in
 * contrastboundary_amd/ may import, link or call it.
 *
 * Pinning: the reference has no CPU pointops and its .cu files need the CUDA toolkit headers
 * (cuda_runtime_api.h via ATen/cuda/CUDAContext.h), which this image lacks, so they cannot be built
 * as oracle/_ref.  This file is pinned by tests/golden/pointops_*.npz — outputs of the reference
 * kernel BODIES executed on the host in the build container by tests/golden/gen_pointops_goldens.py
 * (which reads the .cu files where they lie and never copies them) — and by the known-answer
 * vectors of SURVEY.md §7 (hard part 1).
 *
 * Floating point: compiled with -ffp-contract=off; every expression keeps the reference's
 * association, e.g. d2 = (dx*dx + dy*dy) + dz*dz (knnquery_cuda_kernel.cu:99).
 *
 * Build: make -C oracle   ->  oracle/_build/liboracle.so
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------------------------------------
 * K1 knnquery — knnquery_cuda_kernel.cu:65-111
 * ------------------------------------------------------------------------------------------- */

/* sift the root of a max-heap of size `len` down. Follows reheap(), knnquery_cuda_kernel.cu:21-36:
 * the right child is preferred only when strictly larger (:27); descent stops only when the parent
 * is strictly larger than the chosen child (:29), i.e. equal keys are swapped. */
static void heap_sift_root(float* key, int* val, int len)
{
    int parent = 0;
    for (;;) {
        int kid = 2 * parent + 1;
        if (kid >= len) break;
        if (kid + 1 < len && key[kid + 1] > key[kid]) kid += 1;
        if (key[parent] > key[kid]) break;
        float fk = key[parent]; key[parent] = key[kid]; key[kid] = fk;
        int   iv = val[parent]; val[parent] = val[kid]; val[kid] = iv;
        parent = kid;
    }
}

/* cloud of a stacked row index: first c with row < offset[c] (get_bt_idx, :51-62) */
static int cloud_of(int row, const int* offset)
{
    int c = 0;
    while (row >= offset[c]) c++;
    return c;
}

static void knn_one_query(int q, int nsample, const float* xyz, const float* new_xyz,
                          const int* offset, const int* new_offset, int* idx, float* dist2,
                          float* key, int* val)
{
    const int c = cloud_of(q, new_offset);
    const int lo = (c == 0) ? 0 : offset[c - 1];          /* :75-79 */
    const int hi = offset[c];                              /* :80 */
    const float qx = new_xyz[3 * q + 0], qy = new_xyz[3 * q + 1], qz = new_xyz[3 * q + 2];

    for (int j = 0; j < nsample; j++) { key[j] = 1e10f; val[j] = lo; }   /* :91-94 */
    for (int i = lo; i < hi; i++) {                                       /* :95-105 */
        const float dx = qx - xyz[3 * i + 0];
        const float dy = qy - xyz[3 * i + 1];
        const float dz = qz - xyz[3 * i + 2];
        const float d2 = (dx * dx + dy * dy) + dz * dz;                   /* :99 */
        if (d2 < key[0]) {                                                /* strict, :100 */
            key[0] = d2; val[0] = i;
            heap_sift_root(key, val, nsample);
        }
    }
    /* heap_sort(), :39-48 — ascending */
    for (int last = nsample - 1; last > 0; last--) {
        float fk = key[0]; key[0] = key[last]; key[last] = fk;
        int   iv = val[0]; val[0] = val[last]; val[last] = iv;
        heap_sift_root(key, val, last);
    }
    for (int j = 0; j < nsample; j++) { idx[(size_t)q * nsample + j] = val[j]; dist2[(size_t)q * nsample + j] = key[j]; }
}

/* queries [q0,q1) only — lets the caller time a bounded sample / thread over queries */
ORACLE_API void oracle_knnquery_range(int q0, int q1, int nsample, const float* xyz, const float* new_xyz,
                                      const int* offset, const int* new_offset, int* idx, float* dist2)
{
    float* key = (float*)malloc(sizeof(float) * (size_t)nsample);
    int*   val = (int*)malloc(sizeof(int) * (size_t)nsample);
    for (int q = q0; q < q1; q++) knn_one_query(q, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, key, val);
    free(key); free(val);
}

/* all queries, OpenMP schedule(static) over queries with `threads` threads (<= 0: the runtime's default): the CPU baseline of
 * bench.py at all cores; same per-query function, so the result does not depend on the thread count */
ORACLE_API void oracle_knnquery_omp(int m, int nsample, const float* xyz, const float* new_xyz,
                                    const int* offset, const int* new_offset, int* idx, float* dist2, int threads)
{
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
    {
        float* key = (float*)malloc(sizeof(float) * (size_t)nsample);
        int*   val = (int*)malloc(sizeof(int) * (size_t)nsample);
#pragma omp for schedule(static)
        for (int q = 0; q < m; q++) knn_one_query(q, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, key, val);
        free(key); free(val);
    }
}

ORACLE_API void oracle_knnquery(int m, int nsample, const float* xyz, const float* new_xyz,
                                const int* offset, const int* new_offset, int* idx, float* dist2)
{
    oracle_knnquery_range(0, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2);
}

/* ---------------------------------------------------------------------------------------------
 * K2 furthestsampling — sampling_cuda_kernel.cu:14-129 (+ block size rule cuda_utils.h:11-14)
 *
 * The reference runs one block of B = 2^floor(log2 n_max) (clipped to [1,1024]) threads per cloud.
 * Thread t owns points k = start+t, start+t+B, ... and keeps its FIRST maximum of
 * d2 = min(d(k, last), tmp[k]) (strict '>', :57-58).  The shared-memory tree (:64-123, strides
 * B/2..1, __update keeps the lower slot on ties, :5-10) therefore returns, among the threads tied
 * at the global maximum, the one whose B-bit-reversed thread id is smallest.  This function
 * restates exactly that: per-thread first max, then a lexicographic winner on
 * (larger d2, smaller bitrev(t)).  Threads that own no point carry (-1, start) as in :44-45.
 * ------------------------------------------------------------------------------------------- */
static int ref_block_threads(int n_max)
{
    /* cuda_utils.h:11-14: pow_2 = (int)(log(double(n)) / log(2.0)); clip 1<<pow_2 to [1,1024].
     * Restated with the same double arithmetic; on glibc this equals floor(log2 n) for every
     * n < 2^21 (enumerated in tests/test_oracle_pointops.py). */
    if (n_max < 1) n_max = 1;
    const int p = (int)(log((double)n_max) / log(2.0));
    int t = (p >= 31) ? 1024 : (1 << p);
    if (t > 1024) t = 1024;
    if (t < 1) t = 1;
    return t;
}

static unsigned bitrev(unsigned v, int bits)
{
    unsigned r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}

ORACLE_API int oracle_ref_block_threads(int n_max) { return ref_block_threads(n_max); }

ORACLE_API void oracle_furthestsampling(int b, int n_max, const float* xyz, const int* offset,
                                        const int* new_offset, float* tmp, int* idx)
{
    const int B = ref_block_threads(n_max);
    int bits = 0; while ((1 << bits) < B) bits++;
    float* tbest = (float*)malloc(sizeof(float) * (size_t)B);
    int*   targ  = (int*)malloc(sizeof(int) * (size_t)B);

    for (int c = 0; c < b; c++) {
        const int n0 = (c == 0) ? 0 : offset[c - 1], n1 = offset[c];
        const int m0 = (c == 0) ? 0 : new_offset[c - 1], m1 = new_offset[c];
        int last = n0;                                   /* :26 / :34 */
        if (m1 <= m0) continue;                          /* the reference would still write idx[m0] (:39),
                                                            i.e. into the next cloud's slot; skipped here */
        idx[m0] = n0;                                    /* :39 */
        for (int j = m0 + 1; j < m1; j++) {
            const float lx = xyz[3 * last + 0], ly = xyz[3 * last + 1], lz = xyz[3 * last + 2];
            for (int t = 0; t < B; t++) { tbest[t] = -1.0f; targ[t] = n0; }
            for (int k = n0; k < n1; k++) {
                const int t = (k - n0) % B;
                const float dx = xyz[3 * k + 0] - lx, dy = xyz[3 * k + 1] - ly, dz = xyz[3 * k + 2] - lz;
                const float d = (dx * dx + dy * dy) + dz * dz;               /* :54 */
                const float d2 = fminf(d, tmp[k]);                            /* :55 */
                tmp[k] = d2;
                if (d2 > tbest[t]) { tbest[t] = d2; targ[t] = k; }            /* :57-58 */
            }
            int win = 0; unsigned winrev = bitrev(0u, bits);
            for (int t = 1; t < B; t++) {
                const unsigned r = bitrev((unsigned)t, bits);
                if (tbest[t] > tbest[win] || (tbest[t] == tbest[win] && r < winrev)) { win = t; winrev = r; }
            }
            last = targ[win];                                                 /* :125 */
            idx[j] = last;
        }
    }
    free(tbest); free(targ);
}

/* literal tree version of the reduction, used by the tests to cross-check the bit-reversal claim */
ORACLE_API int oracle_fps_tree_winner(int B, const float* best, const int* arg)
{
    float* v = (float*)malloc(sizeof(float) * (size_t)B);
    int*   a = (int*)malloc(sizeof(int) * (size_t)B);
    memcpy(v, best, sizeof(float) * (size_t)B); memcpy(a, arg, sizeof(int) * (size_t)B);
    for (int s = B / 2; s >= 1; s /= 2)
        for (int t = 0; t < s; t++) {                    /* __update(t, t+s), :5-10 */
            const float v1 = v[t], v2 = v[t + s];
            const int i1 = a[t], i2 = a[t + s];
            v[t] = (v1 > v2) ? v1 : v2;                  /* max() */
            a[t] = (v2 > v1) ? i2 : i1;
        }
    const int r = a[0];
    free(v); free(a);
    return r;
}

/* ---------------------------------------------------------------------------------------------
 * K3/K4 grouping — grouping_cuda_kernel.cu:5-25
 * ------------------------------------------------------------------------------------------- */
ORACLE_API void oracle_grouping_forward(int m, int nsample, int c, const float* input, const int* idx, float* output)
{
    for (size_t r = 0; r < (size_t)m * nsample; r++) {
        const float* src = input + (size_t)idx[r] * c;
        memcpy(output + r * c, src, sizeof(float) * (size_t)c);
    }
}

ORACLE_API void oracle_grouping_backward(int m, int nsample, int c, const float* grad_output, const int* idx, float* grad_input)
{
    for (size_t r = 0; r < (size_t)m * nsample; r++) {
        float* dst = grad_input + (size_t)idx[r] * c;
        for (int ch = 0; ch < c; ch++) dst[ch] += grad_output[r * c + ch];
    }
}

/* ---------------------------------------------------------------------------------------------
 * K5/K6 interpolation — interpolation_cuda_kernel.cu:5-33 (accumulates, neighbours in order)
 * ------------------------------------------------------------------------------------------- */
ORACLE_API void oracle_interpolation_forward(int n, int c, int k, const float* input, const int* idx, const float* weight, float* output)
{
    for (int p = 0; p < n; p++)
        for (int ch = 0; ch < c; ch++)
            for (int i = 0; i < k; i++)
                output[(size_t)p * c + ch] += input[(size_t)idx[(size_t)p * k + i] * c + ch] * weight[(size_t)p * k + i];
}

ORACLE_API void oracle_interpolation_backward(int n, int c, int k, const float* grad_output, const int* idx, const float* weight, float* grad_input)
{
    for (int p = 0; p < n; p++)
        for (int ch = 0; ch < c; ch++)
            for (int i = 0; i < k; i++)
                grad_input[(size_t)idx[(size_t)p * k + i] * c + ch] += grad_output[(size_t)p * c + ch] * weight[(size_t)p * k + i];
}

/* ---------------------------------------------------------------------------------------------
 * K7/K8 subtraction — subtraction_cuda_kernel.cu:5-30
 * ------------------------------------------------------------------------------------------- */
ORACLE_API void oracle_subtraction_forward(int n, int nsample, int c, const float* input1, const float* input2, const int* idx, float* output)
{
    for (int p = 0; p < n; p++)
        for (int s = 0; s < nsample; s++) {
            const size_t r = (size_t)p * nsample + s;
            for (int ch = 0; ch < c; ch++)
                output[r * c + ch] = input1[(size_t)p * c + ch] - input2[(size_t)idx[r] * c + ch];
        }
}

ORACLE_API void oracle_subtraction_backward(int n, int nsample, int c, const int* idx, const float* grad_output, float* grad_input1, float* grad_input2)
{
    for (int p = 0; p < n; p++)
        for (int s = 0; s < nsample; s++) {
            const size_t r = (size_t)p * nsample + s;
            for (int ch = 0; ch < c; ch++) {
                grad_input1[(size_t)p * c + ch] += grad_output[r * c + ch];
                grad_input2[(size_t)idx[r] * c + ch] += -grad_output[r * c + ch];
            }
        }
}

/* ---------------------------------------------------------------------------------------------
 * K9/K10 aggregation — aggregation_cuda_kernel.cu:5-39
 * ------------------------------------------------------------------------------------------- */
ORACLE_API void oracle_aggregation_forward(int n, int nsample, int c, int w_c, const float* input, const float* position,
                                           const float* weight, const int* idx, float* output)
{
    for (int p = 0; p < n; p++)
        for (int ch = 0; ch < c; ch++) {
            const int wch = ch % w_c;
            for (int s = 0; s < nsample; s++) {
                const size_t r = (size_t)p * nsample + s;
                output[(size_t)p * c + ch] += (input[(size_t)idx[r] * c + ch] + position[r * c + ch]) * weight[r * w_c + wch];
            }
        }
}

ORACLE_API void oracle_aggregation_backward(int n, int nsample, int c, int w_c, const float* input, const float* position,
                                            const float* weight, const int* idx, const float* grad_output,
                                            float* grad_input, float* grad_position, float* grad_weight)
{
    for (int p = 0; p < n; p++)
        for (int ch = 0; ch < c; ch++) {
            const int wch = ch % w_c;
            const float g = grad_output[(size_t)p * c + ch];
            for (int s = 0; s < nsample; s++) {
                const size_t r = (size_t)p * nsample + s;
                grad_input[(size_t)idx[r] * c + ch] += g * weight[r * w_c + wch];
                grad_position[r * c + ch] = g * weight[r * w_c + wch];
                grad_weight[r * w_c + wch] += g * (input[(size_t)idx[r] * c + ch] + position[r * c + ch]);
            }
        }
}
