// K1 for 64 < nsample <= 1024 (a7: sub-scene labels ask for the kr = 64 .. 256 nearest stage-0 points of every coarse point):
// one 256-lane workgroup per query, selection instead of the reference's 8 KB per-thread heap
// (/root/reference/pytorch/lib/pointops/src/knnquery/knnquery_cuda_kernel.cu:65-111; call site basic_operators.py:22-30).
//
//   1. bound: the R-th smallest distance among the first 4096 supports of the query's cloud (R = twice the expected share of the
//      K nearest in that sample + 32), by bisection on the float bit patterns with block-wide counts;
//   2. collect: one pass over the cloud appends every support with d2 <= bound to an LDS list (expected ~2K + 320 entries, capacity 2048);
//   3. select: the exact K-th smallest of the list (bisection again), then CERTIFY as the grid kernel does: the K-th value is not
//      shared with a (K+1)-th support and — unless the caller only needs the set — all K distances are distinct.  Then the
//      reference's heap output is the ascending sort of that set, whatever its visiting order was;
//   4. sort the K winners by (d2, index) with a bitonic network in LDS and write the row.
// Anything else (list overflow, fewer than K supports, a tie that matters) goes to the worklist and is redone by the exact kernel
// in the reference's order.  The brute-force wave kernel needs ~1 us per heap insertion on one wave (1.9 ms for 160 queries x 40960
// supports at K = 256); this path reads the cloud ~1.1 times per query and does the rest in LDS.
#include "cbl_common.h"

int cbl_knn_exact_worklist(int b, int m, int K, const float* xyz, const float* new_xyz, const int* offset, const int* new_offset,
                           int* idx, float* dist2, const int* worklist, const int* worklist_count, int max_work, hipStream_t st,
                           const void* grids, const int* cell_start, const void* sorted);   // knn_exact.hip

namespace {

constexpr int SB = 256, SWAVES = SB / 64;
constexpr int S_SAMPLE = 4096, S_PER = S_SAMPLE / SB;
constexpr int CAP = 2048, C_PER = CAP / SB;
constexpr int KMAX = 1024;

struct BlockCount {                      // block-wide sum of per-lane counts; double-buffered so one barrier per call suffices
    int (*buf)[SWAVES];
    int turn;
    __device__ __forceinline__ int sum(int wave_total, int lane, int wave)
    {
        if (lane == 0) buf[turn][wave] = wave_total;
        __syncthreads();
        const int t = buf[turn][0] + buf[turn][1] + buf[turn][2] + buf[turn][3];
        turn ^= 1;
        return t;
    }
};

// smallest bit pattern u with #{v <= u} >= rank, over PER registers per lane (all patterns are of non-negative floats or 0xffffffff)
template <int PER>
__device__ __forceinline__ unsigned kth_smallest_bits(const unsigned (&v)[PER], int rank, BlockCount& bc, int lane, int wave)
{
    unsigned lo = 0u, hi = 0x7f800000u;
    while (lo < hi) {
        const unsigned mid = lo + ((hi - lo) >> 1);
        int c = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) c += __popcll(__ballot(v[j] <= mid));
        if (bc.sum(c, lane, wave) >= rank) hi = mid; else lo = mid + 1;
    }
    return lo;
}

__global__ __launch_bounds__(SB) void knn_select_kernel(int b, int m, int K, const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                        const int* __restrict__ offset, const int* __restrict__ new_offset,
                                                        int* __restrict__ idx, float* __restrict__ dist2,
                                                        int* __restrict__ worklist, int* __restrict__ counters, int set_exact)
{
    __shared__ float cd[CAP];
    __shared__ int ci[CAP];
    __shared__ float sd[KMAX];
    __shared__ int si[KMAX];
    __shared__ int cbuf[2][SWAVES];
    __shared__ int lcount, wcount, dupflag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    BlockCount bc{cbuf, 0};

    for (int q = blockIdx.x; q < m; q += gridDim.x) {
        __syncthreads();                                            // the previous query's LDS lists are no longer read
        const int c = cbl_cloud_of(q, new_offset, b);
        const int start = (c == 0) ? 0 : offset[c - 1], end = offset[c];
        const int n_c = end - start;
        const float qx = new_xyz[3 * q], qy = new_xyz[3 * q + 1], qz = new_xyz[3 * q + 2];
        if (tid == 0) { lcount = 0; wcount = 0; dupflag = 0; }

        // ---- 1. bound from a sample (clouds that fit the list are taken whole)
        float tau0 = INFINITY;
        if (n_c > CAP) {
            const int S = min(n_c, S_SAMPLE);
            const int R = min(S, (int)((long long)K * S * 2 / n_c) + 32);
            unsigned v[S_PER];
#pragma unroll
            for (int j = 0; j < S_PER; j++) {
                const int i = tid + SB * j;
                const int ic = start + min(i, S - 1);
                const float d = cbl_dist2(qx, qy, qz, xyz[3 * ic], xyz[3 * ic + 1], xyz[3 * ic + 2]);
                v[j] = (i < S) ? __float_as_uint(d) : 0xffffffffu;
            }
            tau0 = __uint_as_float(kth_smallest_bits(v, R, bc, lane, wave));
        }
        __syncthreads();

        // ---- 2. collect d2 <= tau0
        constexpr int U = 4;
        for (int base = start; base < end; base += SB * U) {
            float d[U]; int id[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                id[u] = base + u * SB + tid;
                const int ic = min(id[u], end - 1);
                d[u] = cbl_dist2(qx, qy, qz, xyz[3 * ic], xyz[3 * ic + 1], xyz[3 * ic + 2]);      // (new - x)^2 ..., :99
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const bool keep = id[u] < end && d[u] <= tau0;
                const unsigned long long mk = __ballot(keep);
                if (mk) {
                    int off = 0;
                    if (lane == 0) off = atomicAdd(&lcount, __popcll(mk));
                    off = __builtin_amdgcn_readfirstlane(off);
                    const int pos = off + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0));
                    if (keep && pos < CAP) { cd[pos] = d[u]; ci[pos] = id[u]; }
                }
            }
        }
        __syncthreads();
        const int L = lcount;
        bool ok = (L >= K) && (L <= CAP);

        // ---- 3. exact K-th smallest of the list, boundary certification
        unsigned w[C_PER];
#pragma unroll
        for (int j = 0; j < C_PER; j++) { const int e = tid + SB * j; w[j] = (ok && e < L) ? __float_as_uint(cd[e]) : 0xffffffffu; }
        unsigned kth = 0u;
        if (ok) {                                                    // block-uniform
            kth = kth_smallest_bits(w, K, bc, lane, wave);
            int cle = 0;
#pragma unroll
            for (int j = 0; j < C_PER; j++) cle += __popcll(__ballot(w[j] <= kth));
            const int n_le = bc.sum(cle, lane, wave);
            ok = (n_le == K) || set_exact == 2;                      // else the K-th distance is shared with a (K+1)-th support
            if (ok && n_le != K) {                                   // "any tie" policy: of the supports AT the K-th distance the smallest indices stay
                int clt = 0;
#pragma unroll
                for (int j = 0; j < C_PER; j++) clt += __popcll(__ballot(w[j] < kth));
                const int need_eq = K - bc.sum(clt, lane, wave);
                unsigned ti[C_PER];                                  // index of a tied entry, 0xffffffff otherwise
#pragma unroll
                for (int j = 0; j < C_PER; j++) { const int e = tid + SB * j; ti[j] = (w[j] == kth) ? (unsigned)ci[e] : 0xffffffffu; }
                unsigned lo_i = 0u, hi_i = 0x7fffffffu;              // smallest index bound holding need_eq tied entries
                while (lo_i < hi_i) {
                    const unsigned mid = lo_i + ((hi_i - lo_i) >> 1);
                    int cc = 0;
#pragma unroll
                    for (int j = 0; j < C_PER; j++) cc += __popcll(__ballot(ti[j] <= mid));
                    if (bc.sum(cc, lane, wave) >= need_eq) hi_i = mid; else lo_i = mid + 1;
                }
#pragma unroll
                for (int j = 0; j < C_PER; j++) if (w[j] == kth && ti[j] > lo_i) w[j] = 0xffffffffu;     // surplus ties drop out
            }
        }

        // ---- 4. the K winners, sorted by (d2, index)
        int P = 1; while (P < K) P <<= 1;
        if (ok) {
            for (int e = tid; e < P; e += SB) { sd[e] = INFINITY; si[e] = 0x7fffffff; }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < C_PER; j++) {
                const int e = tid + SB * j;
                const bool win = w[j] <= kth;
                const unsigned long long mk = __ballot(win);
                if (mk) {
                    int off = 0;
                    if (lane == 0) off = atomicAdd(&wcount, __popcll(mk));
                    off = __builtin_amdgcn_readfirstlane(off);
                    const int pos = off + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0));
                    if (win) { sd[pos] = cd[e]; si[pos] = ci[e]; }
                }
            }
            __syncthreads();
            for (int k2 = 2; k2 <= P; k2 <<= 1) {
                for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                    for (int t = tid; t < P / 2; t += SB) {
                        const int lo_i = 2 * t - (t & (j2 - 1));                  // pair (lo_i, lo_i + j2)
                        const int hi_i = lo_i + j2;
                        const bool asc = (lo_i & k2) == 0;
                        const float a = sd[lo_i], bb = sd[hi_i]; const int ia = si[lo_i], ib = si[hi_i];
                        const bool a_gt = a > bb || (a == bb && ia > ib);
                        if (a_gt == asc) { sd[lo_i] = bb; sd[hi_i] = a; si[lo_i] = ib; si[hi_i] = ia; }
                    }
                    __syncthreads();
                }
            }
            if (set_exact == 0) {                                    // equal distances inside the list: the reference's order is its heap's
                for (int e = tid; e + 1 < K; e += SB) if (sd[e] == sd[e + 1]) dupflag = 1;
                __syncthreads();
                ok = dupflag == 0;
            } else if (set_exact == 1) {                             // set policy: a tie for column 0 still takes the reference's order (knn_grid.hip)
                if (tid == 0 && K > 1 && sd[0] == sd[1]) dupflag = 1;
                __syncthreads();
                ok = dupflag == 0;
            }
        }
        if (ok) {
            for (int e = tid; e < K; e += SB) { idx[(size_t)q * K + e] = si[e]; dist2[(size_t)q * K + e] = sd[e]; }
        } else if (tid == 0) {
            worklist[atomicAdd(counters, 1)] = q;
        }
    }
}

__global__ void select_init_kernel(int* counters) { if (threadIdx.x == 0) counters[0] = 0; }

}  // namespace

size_t cbl_knn_select_workspace_bytes(int b, int n, int m, int nsample)
{
    if (nsample <= 64 || nsample > KMAX || n < 4096 || b <= 0 || m <= 0) return 0;
    return 256 + sizeof(int) * (size_t)m;
}

int cbl_knn_select_launch(int b, int n, int m, int nsample, const float* xyz, const float* new_xyz, const int* offset, const int* new_offset,
                          int* idx, float* dist2, void* ws, size_t ws_bytes, int set_exact, hipStream_t st)
{
    (void)n;
    if (ws_bytes < cbl_knn_select_workspace_bytes(b, n, m, nsample)) return CBL_ERR_WORKSPACE;
    int* counters = reinterpret_cast<int*>(ws);
    int* worklist = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + 256);
    hipLaunchKernelGGL(select_init_kernel, dim3(1), dim3(64), 0, st, counters);
    hipLaunchKernelGGL(knn_select_kernel, dim3(min(m, 256 * 8)), dim3(SB), 0, st, b, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2,
                       worklist, counters, set_exact);
    int rc = cbl_status();
    if (rc) return rc;
    return cbl_knn_exact_worklist(b, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, worklist, counters, m, st, nullptr, nullptr, nullptr);
}
