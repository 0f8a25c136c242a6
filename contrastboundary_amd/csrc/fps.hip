// K2: furthest point sampling, one 1024-lane workgroup (16 waves) per cloud.
// Replaces furthestsampling_cuda_kernel  /root/reference/pytorch/lib/pointops/src/sampling/sampling_cuda_kernel.cu:14-129.
//
// Same sequence of samples as the reference, including ties: the reference's result depends on its block size
// B = opt_n_threads(n_max) (cuda_utils.h:11-14) through (a) which thread owns a point (t = (k-start) mod B, first maximum wins
// inside a thread, :57-58) and (b) the shared-memory tree (:64-123), which among tied threads returns the smallest BIT-REVERSED
// thread id.  Here that rule is an explicit 64-bit key per point, independent of the real workgroup size:
//     key = (bits(d2) << 32) | ~((bitrev_B(t) << 22) | (k-start)/B)         (d2 >= 0 so the bit pattern orders like the value)
// and every iteration is "update running min distance, max-reduce the key".
//
// MI355X mapping.  FPS is latency bound: m sequential samples, each needing every point's distance to the previous sample.
//  * The whole per-cloud state stays ON CHIP for clouds up to 40960 points: running distances in VGPRs (40 per lane), coordinates
//    in VGPRs for the first rows, in LDS (13 rows x 1024 lanes x 12 B = 156 KiB of the CU's 160 KiB) for the next, and only the
//    last rows are re-read from L2 each sample — issued at the top of the iteration so they land while the resident rows are
//    processed.  (The reference round-trips tmp[] and xyz through global memory every iteration.)
//  * The winner's COORDINATES travel with its key through the reduction (wave max by 6 cross-lane steps, one LDS slot per wave,
//    one barrier, double-buffered), so the next iteration does not start with a dependent global load of xyz[winner].
//  * One barrier per sample instead of the reference's 11, and no read-after-write race on the winner (:125 vs :60-61).
#include "cbl_common.h"
#include "fps_wave.h"
#include <math.h>

namespace {

constexpr int FPS_BLOCK = 1024;
constexpr int FPS_WAVES = FPS_BLOCK / 64;

struct FpsSlot { unsigned long long key; float x, y, z, pad; };
struct FpsCloud { int n0, n1, m0, m1; };

__device__ __forceinline__ FpsCloud fps_cloud(const int* __restrict__ offset, const int* __restrict__ new_offset)
{
    const int c = blockIdx.x;
    FpsCloud r;
    r.n0 = c ? offset[c - 1] : 0;       r.n1 = offset[c];
    r.m0 = c ? new_offset[c - 1] : 0;   r.m1 = new_offset[c];
    return r;
}

// rank of local point kk under reference block size 2^bits: smaller = preferred among equal d2
__device__ __forceinline__ unsigned fps_rank(int kk, int bits)
{
    const unsigned t = (unsigned)kk & ((1u << bits) - 1u);
    const unsigned j = (unsigned)kk >> bits;
    const unsigned rev = bits ? (__brev(t) >> (32 - bits)) : 0u;
    return (rev << 22) | j;
}
__device__ __forceinline__ int fps_unrank(unsigned rank, int bits)
{
    const unsigned rev = rank >> 22, j = rank & ((1u << 22) - 1u);
    const unsigned t = bits ? (__brev(rev) >> (32 - bits)) : 0u;
    return (int)((j << bits) | t);
}

// Running best of one lane.  With a 1024-lane workgroup every point a lane owns has the same reference thread id (B = 1024:
// t = lane; B < 1024 means n_max < 1024, i.e. a single row), so inside a lane the rank order is the row order and the
// reference's "first maximum wins" (:57-58) is a strict '>' in row order: one compare and two selects per point, no branch.
struct FpsBest {
    float d; int row;
    __device__ __forceinline__ void init() { d = -1.f; row = 0; }
    __device__ __forceinline__ void offer(float d2, int jj) { const bool up = d2 > d; d = up ? d2 : d; row = up ? jj : row; }
};

// ---- cross-lane reductions: fps_wave.h (row_max_f, row_min_u, wave_max_f, wave_min_u)

// Block-wide winner of the lexicographic (larger d2, smaller rank) maximum, with its coordinates.  `coords(row, x, y, z)` is
// called by ONE lane per wave (the wave's winner) to fetch the coordinates of its best row.  One barrier; slots[par] alternates.
template <class Coords>
__device__ __forceinline__ void block_winner(const FpsBest& b, bool any, int tid, FpsSlot (*slots)[FPS_WAVES], int par, int bits,
                                             Coords&& coords, int& win_local, float& wx, float& wy, float& wz)
{
    const int lane = tid & 63, wave = tid >> 6;
    const float d = any ? b.d : -3.f;
    const unsigned rank = any ? fps_rank(tid + b.row * FPS_BLOCK, bits) : 0xffffffffu;
    const float wd = wave_max_f(d);
    const unsigned wr = wave_min_u(d == wd ? rank : 0xffffffffu);
    if (d == wd && rank == wr) {                                     // exactly one lane (ranks are unique), or lane(s) of an empty wave
        FpsSlot s; s.key = ((unsigned long long)__float_as_uint(wd) << 32) | wr;
        s.x = s.y = s.z = 0.f; s.pad = 0.f;
        if (any) coords(b.row, s.x, s.y, s.z);
        if (any || lane == 0) slots[par][wave] = s;
    }
    __syncthreads();
    const FpsSlot mine = slots[par][lane & (FPS_WAVES - 1)];
    const float sd = __uint_as_float((unsigned)(mine.key >> 32)); const unsigned sr = (unsigned)(mine.key & 0xffffffffu);
    const float bd = row_max_f(sd);
    const unsigned br = row_min_u(sd == bd ? sr : 0xffffffffu);
    const int slot = __builtin_ctzll(__ballot(sd == bd && sr == br) & 0xffffull);
    wx = slots[par][slot].x; wy = slots[par][slot].y; wz = slots[par][slot].z;
    win_local = fps_unrank(br, bits);
}

// PER rows of 1024 points per cloud.  Rows [0,RREG): xyz in VGPRs; [RREG, RREG+RLDS): xyz in LDS; the rest re-read from L2.
// Lanes past the end of the (single) partial row hold a clamped valid point and a running distance of -2, which can never
// win the max, so the sample loop needs no per-lane predicates; absent rows are skipped wave-uniformly.
template <int PER, int RREG, int RLDS>
__global__ __launch_bounds__(FPS_BLOCK) void fps_kernel(int bits, const float* __restrict__ xyz,
                                                        const int* __restrict__ offset, const int* __restrict__ new_offset,
                                                        float* __restrict__ tmp, int* __restrict__ idx, const int* __restrict__ prefix_cert)
{
    constexpr int RGLB = PER - RREG - RLDS;                         // rows streamed from L2, in batches of <= GBATCH rows
    constexpr int GBATCH = 7;
    constexpr int NB = (RGLB + GBATCH - 1) / GBATCH;
    __shared__ FpsSlot slots[2][FPS_WAVES];
    extern __shared__ __attribute__((aligned(16))) float lds_xyz[];     // [3][RLDS][1024]
    const FpsCloud cl = fps_cloud(offset, new_offset);
    if (cl.m1 <= cl.m0) return;
    if (prefix_cert && prefix_cert[blockIdx.x] >= cl.m1 - cl.m0) return;       // this cloud's samples are a certified prefix (fps_prefix_kernel wrote them)
    const int tid = threadIdx.x, nloc = cl.n1 - cl.n0;
    const float* __restrict__ P = xyz + (size_t)3 * cl.n0;
    const int omax = 3 * (nloc - 1);

    float t[PER];
    float rx[RREG > 0 ? RREG : 1], ry[RREG > 0 ? RREG : 1], rz[RREG > 0 ? RREG : 1];
#pragma unroll
    for (int jj = 0; jj < PER; jj++) {
        const int kk = tid + jj * FPS_BLOCK;
        t[jj] = (kk < nloc) ? tmp[cl.n0 + kk] : -2.f;
        const int o = min(3 * kk, omax);
        if (jj < RREG) { rx[jj] = P[o]; ry[jj] = P[o + 1]; rz[jj] = P[o + 2]; }
        else if (jj < RREG + RLDS) {
            const int l = (jj - RREG) * FPS_BLOCK + tid;
            lds_xyz[l] = P[o]; lds_xyz[RLDS * FPS_BLOCK + l] = P[o + 1]; lds_xyz[2 * RLDS * FPS_BLOCK + l] = P[o + 2];
        }
    }
    if (tid == 0) idx[cl.m0] = cl.n0;                                // :39
    float lx = P[0], ly = P[1], lz = P[2];                           // first sample = first point of the cloud (:26 / :34)

    // coordinates of row `row` of this lane, wherever they live (called by one lane per wave and sample)
    auto coords = [&](int row, float& x, float& y, float& z) {
        if (row < RREG) {
#pragma unroll
            for (int jj = 0; jj < RREG; jj++) if (jj == row) { x = rx[jj]; y = ry[jj]; z = rz[jj]; }
        } else if (row < RREG + RLDS) {
            const int l = (row - RREG) * FPS_BLOCK + tid;
            x = lds_xyz[l]; y = lds_xyz[RLDS * FPS_BLOCK + l]; z = lds_xyz[2 * RLDS * FPS_BLOCK + l];
        } else {
            const int o = min(3 * (tid + row * FPS_BLOCK), omax);
            x = P[o]; y = P[o + 1]; z = P[o + 2];
        }
    };

    for (int j = cl.m0 + 1; j < cl.m1; j++) {
        FpsBest b; b.init();
        // opaque per-iteration copy of the lane's element offset: stops hipcc from hoisting loop-invariant 64-bit addresses of
        // every streamed row out of the sample loop (that alone spilled > 100 VGPRs)
        int eoff = 3 * tid;
        asm volatile("" : "+v"(eoff));
        auto row_off = [&](int jj) { int o = eoff + 3 * jj * FPS_BLOCK; if ((jj + 1) * FPS_BLOCK > nloc) o = min(o, omax); return o; };
        auto update = [&](int jj, float x, float y, float z) {
            const float d = cbl_dist2(x, y, z, lx, ly, lz);          // :54
            const float d2 = fminf(d, t[jj]);                        // :55
            t[jj] = d2;
            b.offer(d2, jj);
        };
        // resident rows are cut into NB+1 slices; batch k of the streamed rows is requested before slice k and consumed after it
        constexpr int RES = RREG + RLDS;
        auto resident = [&](int jj) {
            if (jj * FPS_BLOCK < nloc) {
                if (jj < RREG) update(jj, rx[jj], ry[jj], rz[jj]);
                else { const int l = (jj - RREG) * FPS_BLOCK + tid; update(jj, lds_xyz[l], lds_xyz[RLDS * FPS_BLOCK + l], lds_xyz[2 * RLDS * FPS_BLOCK + l]); }
            }
        };
#pragma unroll
        for (int k = 0; k < NB; k++) {
            float gx[GBATCH], gy[GBATCH], gz[GBATCH];
#pragma unroll
            for (int g = 0; g < GBATCH; g++) {
                const int jj = RES + k * GBATCH + g;
                if (jj < PER && jj * FPS_BLOCK < nloc) { const int o = row_off(jj); gx[g] = P[o]; gy[g] = P[o + 1]; gz[g] = P[o + 2]; }
            }
#pragma unroll
            for (int jj = (RES * k) / (NB + 1); jj < (RES * (k + 1)) / (NB + 1); jj++) resident(jj);
#pragma unroll
            for (int g = 0; g < GBATCH; g++) {
                const int jj = RES + k * GBATCH + g;
                if (jj < PER && jj * FPS_BLOCK < nloc) update(jj, gx[g], gy[g], gz[g]);
            }
            __builtin_amdgcn_sched_barrier(0);                       // keep one batch's loads live at a time
        }
#pragma unroll
        for (int jj = (RES * NB) / (NB + 1); jj < RES; jj++) resident(jj);
        int win;
        block_winner(b, tid < nloc, tid, slots, j & 1, bits, coords, win, lx, ly, lz);
        if (tid == 0) idx[j] = cl.n0 + win;
    }
    // write the running distances back (same side effect as :56).  Slots past the end hold -2; the lane id is laundered so these
    // addresses / masks are not kept alive across the sample loop.
    int tid_o = tid;
    asm volatile("" : "+v"(tid_o));
#pragma unroll
    for (int jj = 0; jj < PER; jj++)
        if (t[jj] >= 0.f) tmp[cl.n0 + tid_o + jj * FPS_BLOCK] = t[jj];
}

// any cloud size: running distances stay in tmp[] (L2-resident), as in the reference
__global__ __launch_bounds__(FPS_BLOCK) void fps_stream_kernel(int bits, const float* __restrict__ xyz,
                                                               const int* __restrict__ offset, const int* __restrict__ new_offset,
                                                               float* __restrict__ tmp, int* __restrict__ idx, const int* __restrict__ prefix_cert)
{
    __shared__ FpsSlot slots[2][FPS_WAVES];
    const FpsCloud cl = fps_cloud(offset, new_offset);
    if (cl.m1 <= cl.m0) return;
    if (prefix_cert && prefix_cert[blockIdx.x] >= cl.m1 - cl.m0) return;
    const int tid = threadIdx.x, nloc = cl.n1 - cl.n0;
    const float* __restrict__ P = xyz + (size_t)3 * cl.n0;
    float* __restrict__ T = tmp + cl.n0;
    if (tid == 0) idx[cl.m0] = cl.n0;
    float lx = P[0], ly = P[1], lz = P[2];
    auto coords = [&](int row, float& x, float& y, float& z) { const int o = 3 * (tid + row * FPS_BLOCK); x = P[o]; y = P[o + 1]; z = P[o + 2]; };
    for (int j = cl.m0 + 1; j < cl.m1; j++) {
        FpsBest b; b.init();
        int row = 0;
        for (int kk = tid; kk < nloc; kk += FPS_BLOCK, row++) {
            const float d2 = fminf(cbl_dist2(P[3 * kk + 0], P[3 * kk + 1], P[3 * kk + 2], lx, ly, lz), T[kk]);
            T[kk] = d2;
            b.offer(d2, row);
        }
        int win;
        block_winner(b, tid < nloc, tid, slots, j & 1, bits, coords, win, lx, ly, lz);
        if (tid == 0) idx[j] = cl.n0 + win;
    }
}

// FPS of an FPS sequence is its prefix.  Let A = (s_0, s_1, ...) be the samples of a cloud P in sampling order.  Sampling A again picks s_0 (the
// first row), and at every later step the point of P that the first run picked — the maximiser over P of the running distance — lies in A and has the
// same running distance there (same samples, same order, same arithmetic, both runs starting from 1e10), so it is the maximiser over A as well, PROVIDED
// the maximum over P was attained by one point only (with ties the reference's rank rule decides, and ranks are positions, which differ between P and A).
// cert[c] = number of leading samples of cloud c that were unique maxima (written by the run that produced A): a request for m <= cert[c] samples of A is
// answered with 0 .. m-1 and the certificate is handed on; anything else runs the sampler.
__global__ __launch_bounds__(256) void fps_prefix_kernel(int b, const int* __restrict__ offset, const int* __restrict__ new_offset,
                                                         const int* __restrict__ cert_in, int* __restrict__ idx, int* __restrict__ cert_out)
{
    const FpsCloud cl = fps_cloud(offset, new_offset);
    const int m = cl.m1 - cl.m0;
    const bool ok = cert_in && m > 0 && cert_in[blockIdx.x] >= m;
    if (ok)
        for (int j = threadIdx.x; j < m; j += 256) idx[cl.m0 + j] = cl.n0 + j;
    if (threadIdx.x == 0 && cert_out) cert_out[blockIdx.x] = ok ? cert_in[blockIdx.x] : 0;     // 0 = no certificate (the sampler may overwrite it)
}

template <int PER, int RREG, int RLDS>
void launch_fps(int b, int bits, const float* xyz, const int* offset, const int* new_offset, float* tmp, int* idx, const int* prefix_cert, hipStream_t st)
{
    const size_t lds = (size_t)3 * RLDS * FPS_BLOCK * sizeof(float);
    if (lds > 48 * 1024) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fps_kernel<PER, RREG, RLDS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
    }
    hipLaunchKernelGGL((fps_kernel<PER, RREG, RLDS>), dim3(b), dim3(FPS_BLOCK), lds, st, bits, xyz, offset, new_offset, tmp, idx, prefix_cert);
}

}  // namespace

// reference block size rule, cuda_utils.h:11-14 (same double arithmetic)
static int ref_block_threads(int n_max)
{
    if (n_max < 1) n_max = 1;
    const int p = (int)(log((double)n_max) / log(2.0));
    int t = (p >= 31) ? 1024 : (1 << p);
    if (t > 1024) t = 1024;
    if (t < 1) t = 1;
    return t;
}

// fps_bucket.hip
size_t cbl_fps_bucket_workspace_bytes(int b, int n);
int cbl_fps_bucket_launch(int b, int n, int n_max, int bits, const float* xyz, const int* offset, const int* new_offset, float* tmp, int* idx,
                          void* ws, size_t ws_bytes, hipStream_t st, const int* prefix_cert, int* cert_out);

// clouds from this size on take the bucket-pruned kernel (same samples): ~1.07 us per sample at any size, against 1.1 (<= 2560
// points) .. 1.5 (10240) .. 6.9 us (40960) for the dense kernels; measured crossover between 2560 and 5000 points
constexpr int FPS_BUCKET_MIN_POINTS = 3072, FPS_BUCKET_MAX_POINTS = 131072;

CBL_EXPORT size_t cbl_furthestsampling_workspace_bytes(int b, int n, int n_max)
{
    if (b <= 0 || n <= 0 || b > 65535 || n_max < FPS_BUCKET_MIN_POINTS || n_max > FPS_BUCKET_MAX_POINTS) return 0;
    return cbl_fps_bucket_workspace_bytes(b, n);
}

static int fps_dense_launch(int b, int n_max, const float* xyz, const int* offset, const int* new_offset, float* tmp, int* idx, const int* prefix_cert,
                            hipStream_t st)
{
    const int B = ref_block_threads(n_max);
    int bits = 0; while ((1 << bits) < B) bits++;
    if (n_max <= 1 * FPS_BLOCK)       launch_fps<1, 1, 0>(b, bits, xyz, offset, new_offset, tmp, idx, prefix_cert, st);
    else if (n_max <= 4 * FPS_BLOCK)  launch_fps<4, 4, 0>(b, bits, xyz, offset, new_offset, tmp, idx, prefix_cert, st);
    else if (n_max <= 10 * FPS_BLOCK) launch_fps<10, 10, 0>(b, bits, xyz, offset, new_offset, tmp, idx, prefix_cert, st);
    else if (n_max <= 16 * FPS_BLOCK) launch_fps<16, 16, 0>(b, bits, xyz, offset, new_offset, tmp, idx, prefix_cert, st);
    else if (n_max <= 27 * FPS_BLOCK) launch_fps<27, 14, 13>(b, bits, xyz, offset, new_offset, tmp, idx, prefix_cert, st);
    else if (n_max <= 40 * FPS_BLOCK) launch_fps<40, 6, 13>(b, bits, xyz, offset, new_offset, tmp, idx, prefix_cert, st);
    else                              hipLaunchKernelGGL(fps_stream_kernel, dim3(b), dim3(FPS_BLOCK), 0, st, bits, xyz, offset, new_offset, tmp, idx, prefix_cert);
    return cbl_status();
}

// the sampler for CHAINS of samplings (the network's four TransitionDown stages each sample the previous stage's samples): see fps_prefix_kernel
CBL_EXPORT int cbl_furthestsampling_chain(int b, int n, int n_max, const float* xyz, const int* offset, const int* new_offset, float* tmp, int* idx,
                                          const int* cert_in, int* cert_out, void* workspace, size_t workspace_bytes, void* stream)
{
    if (b < 0 || n < 0 || n_max < 0) return CBL_ERR_BAD_ARG;
    if (b == 0 || n == 0) return CBL_OK;
    if (!xyz || !offset || !new_offset || !tmp || !idx || !cert_out) return CBL_ERR_BAD_ARG;
    hipStream_t st = cbl_stream(stream);
    hipLaunchKernelGGL(fps_prefix_kernel, dim3(b), dim3(256), 0, st, b, offset, new_offset, cert_in, idx, cert_out);
    const size_t need = cbl_furthestsampling_workspace_bytes(b, n, n_max);
    if (need == 0 || !workspace) return fps_dense_launch(b, n_max, xyz, offset, new_offset, tmp, idx, cert_in, st);
    if (workspace_bytes < need) return CBL_ERR_WORKSPACE;
    const int B = ref_block_threads(n_max);
    int bits = 0; while ((1 << bits) < B) bits++;
    return cbl_fps_bucket_launch(b, n, n_max, bits, xyz, offset, new_offset, tmp, idx, workspace, workspace_bytes, st, cert_in, cert_out);
}

CBL_EXPORT int cbl_furthestsampling_ws(int b, int n, int n_max, const float* xyz, const int* offset, const int* new_offset,
                                       float* tmp, int* idx, void* workspace, size_t workspace_bytes, void* stream)
{
    if (b < 0 || n < 0 || n_max < 0) return CBL_ERR_BAD_ARG;
    if (b == 0 || n == 0) return CBL_OK;
    if (!xyz || !offset || !new_offset || !tmp || !idx) return CBL_ERR_BAD_ARG;
    const size_t need = cbl_furthestsampling_workspace_bytes(b, n, n_max);
    if (need == 0 || !workspace) return cbl_furthestsampling(b, n_max, xyz, offset, new_offset, tmp, idx, stream);
    if (workspace_bytes < need) return CBL_ERR_WORKSPACE;
    const int B = ref_block_threads(n_max);
    int bits = 0; while ((1 << bits) < B) bits++;
    return cbl_fps_bucket_launch(b, n, n_max, bits, xyz, offset, new_offset, tmp, idx, workspace, workspace_bytes, cbl_stream(stream), nullptr, nullptr);
}

CBL_EXPORT int cbl_furthestsampling(int b, int n_max, const float* xyz, const int* offset, const int* new_offset,
                                    float* tmp, int* idx, void* stream)
{
    if (b < 0 || n_max < 0) return CBL_ERR_BAD_ARG;
    if (b == 0) return CBL_OK;
    if (!xyz || !offset || !new_offset || !tmp || !idx) return CBL_ERR_BAD_ARG;
    return fps_dense_launch(b, n_max, xyz, offset, new_offset, tmp, idx, nullptr, cbl_stream(stream));
}
