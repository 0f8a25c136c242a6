/*
 * tfops_oracle.c — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the reference's TF-side CPU ops on the hot path:
 *   N1 grid_subsampling / batch_grid_subsampling   tensorflow/ops/tf_custom_ops/tf_subsampling/grid_subsampling/grid_subsampling.cpp:6-161
 *   N3 grid_subsampling with features + labels      tensorflow/ops/cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-106
 *   N2 batch_nanoflann_neighbors (+ the callers' crop to neighborhood_limits, datasets/base.py:756-765)
 *                                                    tensorflow/ops/tf_custom_ops/tf_neighbors/neighbors/neighbors.cpp:213-336
 *   N4 cpp_knn_batch                                 tensorflow/ops/nearest_neighbors/knn_.cxx:72-135
 * Pinned against oracle/_ref (the reference's own sources compiled where they lie) in tests/test_oracle_tfops.py.
 *
 * Where the reference's output ORDER is implementation-defined, this file fixes a canonical one and the tests compare
 * modulo that freedom:  N1/N3 emit voxels in libstdc++ unordered_map iteration order (:93) -> here ascending voxel key;
 * N2 / N4 order equal distances by std::sort / kd-tree traversal -> here (d2, index).  Barycentre sums are accumulated in
 * INPUT order per voxel exactly like the reference (:76-84), so coordinates are bit-identical.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

typedef struct { uint64_t key; int idx; } KeyIdx;
static int cmp_keyidx(const void* a, const void* b)
{
    const KeyIdx* x = (const KeyIdx*)a; const KeyIdx* y = (const KeyIdx*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

/* one cloud: points [n,3] (+ optional features [n,fdim], labels [n,ldim]) -> voxel barycentres in ascending key order.
 * returns the number of voxels.  label_tie (optional, per output voxel*ldim): 1 if the majority vote was tied. */
static int grid_subsample_cloud(int n, const float* pts, int fdim, const float* feat, int ldim, const int* lab, float dl,
                                float* out_pts, float* out_feat, int* out_lab, int* label_tie)
{
    if (n == 0) return 0;
    float mn[3], mx[3];
    for (int a = 0; a < 3; a++) { mn[a] = pts[a]; mx[a] = pts[a]; }
    for (int i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) { const float v = pts[3 * i + a]; if (v < mn[a]) mn[a] = v; if (v > mx[a]) mx[a] = v; }
    const float inv = 1 / dl;                                           /* (1/sampleDl), :28 */
    float org[3];
    for (int a = 0; a < 3; a++) org[a] = floorf(mn[a] * inv) * dl;     /* floor(minCorner * (1/dl)) * dl, :28 */
    const size_t NX = (size_t)floorf((mx[0] - org[0]) / dl) + 1;       /* :31 */
    const size_t NY = (size_t)floorf((mx[1] - org[1]) / dl) + 1;       /* :32 */
    KeyIdx* ki = (KeyIdx*)malloc(sizeof(KeyIdx) * (size_t)n);
    for (int i = 0; i < n; i++) {
        const size_t iX = (size_t)floorf((pts[3 * i + 0] - org[0]) / dl);  /* :61-63 */
        const size_t iY = (size_t)floorf((pts[3 * i + 1] - org[1]) / dl);
        const size_t iZ = (size_t)floorf((pts[3 * i + 2] - org[2]) / dl);
        ki[i].key = iX + NX * iY + NX * NY * iZ;                       /* :64 */
        ki[i].idx = i;
    }
    qsort(ki, (size_t)n, sizeof(KeyIdx), cmp_keyidx);                   /* by key, input order inside a voxel */
    int m = 0;
    for (int s = 0; s < n;) {
        int e = s;
        float sum[3] = {0, 0, 0};
        while (e < n && ki[e].key == ki[s].key) {                       /* point += p in input order, :76-84 */
            for (int a = 0; a < 3; a++) sum[a] += pts[3 * ki[e].idx + a];
            e++;
        }
        const int count = e - s;
        const float rc = (float)(1.0 / count);                          /* point * (1.0 / count): double -> float arg, :95 */
        for (int a = 0; a < 3; a++) out_pts[3 * m + a] = sum[a] * rc;
        if (fdim > 0 && feat) {
            for (int c = 0; c < fdim; c++) {
                float fs = 0;
                for (int t = s; t < e; t++) fs += feat[(size_t)ki[t].idx * fdim + c];
                out_feat[(size_t)m * fdim + c] = fs / (float)count;     /* f / count, wrapper flavour :88-96 */
            }
        }
        if (ldim > 0 && lab) {
            for (int c = 0; c < ldim; c++) {                            /* majority vote per label column, :97-102 */
                int best = 0, bestcnt = -1, tie = 0;
                for (int t = s; t < e; t++) {
                    const int l = lab[(size_t)ki[t].idx * ldim + c];
                    int cnt = 0;
                    for (int u = s; u < e; u++) cnt += lab[(size_t)ki[u].idx * ldim + c] == l;
                    if (cnt > bestcnt || (cnt == bestcnt && l < best)) { tie = (cnt == bestcnt && l != best) ? 1 : (cnt > bestcnt ? 0 : tie); best = l; bestcnt = cnt; }
                    else if (cnt == bestcnt && l != best) tie = 1;
                }
                out_lab[(size_t)m * ldim + c] = best;                   /* canonical: smallest label among the maxima */
                if (label_tie) label_tie[(size_t)m * ldim + c] = tie;
            }
        }
        m++; s = e;
    }
    free(ki);
    return m;
}

ORACLE_API int oracle_batch_grid_subsampling(int n, const float* points, int b, const int* lengths, float dl,
                                             float* out_points, int* out_lengths)
{
    int start = 0, total = 0;
    for (int c = 0; c < b; c++) {                                       /* clouds one after the other, :133-159 */
        const int m = grid_subsample_cloud(lengths[c], points + 3 * (size_t)start, 0, NULL, 0, NULL, dl, out_points + 3 * (size_t)total, NULL, NULL, NULL);
        out_lengths[c] = m; total += m; start += lengths[c];
    }
    (void)n;
    return total;
}

ORACLE_API int oracle_grid_subsampling_full(int n, const float* points, int fdim, const float* features, int ldim, const int* labels, float dl,
                                            float* out_points, float* out_features, int* out_labels, int* label_tie)
{
    return grid_subsample_cloud(n, points, fdim, features, ldim, labels, dl, out_points, out_features, out_labels, label_tie);
}

typedef struct { float d2; int idx; } DistIdx;
static int cmp_distidx(const void* a, const void* b)
{
    const DistIdx* x = (const DistIdx*)a; const DistIdx* y = (const DistIdx*)b;
    if (x->d2 != y->d2) return x->d2 < y->d2 ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

/* out (nq, limit) global support indices sorted by (d2, idx), padded with ns; counts (nq) = true number within the radius.
 * returns max count (the reference's max_count = its number of output columns before the crop). */
ORACLE_API int oracle_radius_neighbors(int nq, const float* q, int ns, const float* s, int b, const int* q_len, const int* s_len,
                                       float radius, int limit, int* out, int* counts)
{
    const float r2 = radius * radius;                                   /* :230 */
    DistIdx* buf = (DistIdx*)malloc(sizeof(DistIdx) * (size_t)(ns > 0 ? ns : 1));
    int qs = 0, ss = 0, max_count = 0;
    for (int c = 0; c < b; c++) {
        for (int i = qs; i < qs + q_len[c]; i++) {
            int cnt = 0;
            for (int j = ss; j < ss + s_len[c]; j++) {
                const float dx = q[3 * i] - s[3 * j], dy = q[3 * i + 1] - s[3 * j + 1], dz = q[3 * i + 2] - s[3 * j + 2];
                const float d2 = (dx * dx + dy * dy) + dz * dz;         /* nanoflann L2_Simple_Adaptor: sum of diff*diff over dims */
                if (d2 < r2) { buf[cnt].d2 = d2; buf[cnt].idx = j; cnt++; }   /* strict, nanoflann.hpp:249-253 */
            }
            qsort(buf, (size_t)cnt, sizeof(DistIdx), cmp_distidx);      /* sorted = true, :268 */
            for (int k = 0; k < limit; k++) out[(size_t)i * limit + k] = k < cnt ? buf[k].idx : ns;   /* pad with supports.size(), :328 */
            if (counts) counts[i] = cnt;
            if (cnt > max_count) max_count = cnt;
        }
        qs += q_len[c]; ss += s_len[c];
    }
    free(buf);
    return max_count;
}

/* dense batch KNN: points (B,N,3), queries (B,M,3) -> indices (B,M,K) int64 local to each batch element, by (d2, idx) */
ORACLE_API void oracle_knn_batch(int B, int N, int M, int K, const float* pts, const float* queries, long long* out)
{
    DistIdx* buf = (DistIdx*)malloc(sizeof(DistIdx) * (size_t)N);
    for (int bb = 0; bb < B; bb++)
        for (int i = 0; i < M; i++) {
            const float* qp = queries + ((size_t)bb * M + i) * 3;
            for (int j = 0; j < N; j++) {
                const float* p = pts + ((size_t)bb * N + j) * 3;
                const float dx = qp[0] - p[0], dy = qp[1] - p[1], dz = qp[2] - p[2];
                buf[j].d2 = (dx * dx + dy * dy) + dz * dz; buf[j].idx = j;
            }
            qsort(buf, (size_t)N, sizeof(DistIdx), cmp_distidx);
            for (int k = 0; k < K; k++) out[((size_t)bb * M + i) * K + k] = buf[k].idx;
        }
    free(buf);
}
