// K2: furthest point sampling, one 1024-lane workgroup (16 waves) per cloud.
// Replaces furthestsampling_cuda_kernel  /root/reference/pytorch/lib/pointops/src/sampling/sampling_cuda_kernel.cu:14-129.
//
// Same sequence of samples as the reference, including ties: the reference's result depends on its
// block size B = opt_n_threads(n_max) (cuda_utils.h:11-14) through (a) which thread owns a point
// (t = (k-start) mod B, first maximum wins inside a thread, :57-58) and (b) the shared-memory tree
// (:64-123), which among tied threads returns the smallest BIT-REVERSED thread id.  Here that rule is
// an explicit 64-bit key per point, independent of the real workgroup size:
//     key = (bits(d2) << 32) | ~((bitrev_B(t) << 22) | (k-start)/B)         (d2 >= 0 so bits order)
// and every iteration is "update running min distance, max-reduce the key".  MI355X mapping: the
// running distances (and, for clouds <= 16384 points, the coordinates too) stay in VGPRs for the
// whole launch instead of round-tripping through global memory every iteration as the reference's
// tmp[] does; the reduction is 6 cross-lane steps + one LDS exchange between the 16 waves, one
// barrier per sample (double-buffered slots) instead of the reference's 11; the `old` read-after-
// write race of the reference (:125 vs :60-61) does not exist because the winner is recomputed by
// every thread from the exchanged keys.
#include "cbl_common.h"
#include <math.h>

namespace {

constexpr int FPS_BLOCK = 1024;
constexpr int FPS_WAVES = FPS_BLOCK / 64;

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const unsigned long long o = __shfl_xor(v, s);
        v = o > v ? o : v;
    }
    return v;
}

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

// all threads: exchange per-wave maxima through LDS slot set `par`, return the block-wide max key
__device__ __forceinline__ unsigned long long block_max_key(unsigned long long key, unsigned long long (*slots)[FPS_WAVES], int par)
{
    key = wave_max_u64(key);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) slots[par][wave] = key;
    __syncthreads();
    unsigned long long v = slots[par][lane & (FPS_WAVES - 1)];
#pragma unroll
    for (int s = FPS_WAVES / 2; s >= 1; s >>= 1) {
        const unsigned long long o = __shfl_xor(v, s);
        v = o > v ? o : v;
    }
    return v;
}

// PER = points per thread kept in registers (cloud size <= PER*1024); XYZ_REG: coordinates too.
template <int PER, bool XYZ_REG>
__global__ __launch_bounds__(FPS_BLOCK) void fps_reg_kernel(int bits, const float* __restrict__ xyz,
                                                            const int* __restrict__ offset, const int* __restrict__ new_offset,
                                                            float* __restrict__ tmp, int* __restrict__ idx)
{
    __shared__ unsigned long long slots[2][FPS_WAVES];
    const FpsCloud cl = fps_cloud(offset, new_offset);
    if (cl.m1 <= cl.m0) return;
    const int tid = threadIdx.x, nloc = cl.n1 - cl.n0;
    const float* __restrict__ P = xyz + (size_t)3 * cl.n0;

    // Row jj of the cloud = local points [jj*1024, (jj+1)*1024).  Rows are wave-uniformly present or
    // absent; only the last present row can be partial.  Lanes past the end of a partial row get a
    // clamped (valid) address and a running distance of -2, which can never win the max, so the sample
    // loop needs no per-lane predicates (40 loop-invariant lane masks would otherwise eat the SGPRs).
    const int omax = 3 * (nloc - 1);
    float t[PER], px[XYZ_REG ? PER : 1], py[XYZ_REG ? PER : 1], pz[XYZ_REG ? PER : 1];
#pragma unroll
    for (int jj = 0; jj < PER; jj++) {
        const int kk = tid + jj * FPS_BLOCK;
        const bool ok = kk < nloc;
        t[jj] = ok ? tmp[cl.n0 + kk] : -2.f;
        if (XYZ_REG) {
            const int o = min(3 * kk, omax);
            px[jj] = P[o + 0]; py[jj] = P[o + 1]; pz[jj] = P[o + 2];
        }
    }
    if (tid == 0) idx[cl.m0] = cl.n0;                                     // :39
    int last = 0;                                                          // local index of the previous sample
    for (int j = cl.m0 + 1; j < cl.m1; j++) {
        const float lx = P[3 * last + 0], ly = P[3 * last + 1], lz = P[3 * last + 2];   // uniform -> scalar
        float bd = -1.f; unsigned brank = 0xffffffffu;
        // opaque per-iteration copy of the lane's element offset: stops the compiler from hoisting 40
        // loop-invariant 64-bit addresses out of the sample loop (that alone spilled ~130 VGPRs)
        int eoff = 3 * tid;
        asm volatile("" : "+v"(eoff));
#pragma unroll
        for (int jj = 0; jj < PER; jj++) {
            if (jj * FPS_BLOCK < nloc) {                                      // uniform: row present
                float x, y, z;
                if (XYZ_REG) { x = px[jj]; y = py[jj]; z = pz[jj]; }
                else {
                    int o = eoff + 3 * jj * FPS_BLOCK;
                    if ((jj + 1) * FPS_BLOCK > nloc) o = min(o, omax);       // uniform: partial row
                    x = P[o + 0]; y = P[o + 1]; z = P[o + 2];
                }
                const float d = cbl_dist2(x, y, z, lx, ly, lz);              // :54
                const float d2 = fminf(d, t[jj]);                             // :55
                t[jj] = d2;
                if (d2 >= bd) {                                               // candidate for (d2, rank) max
                    const unsigned r = fps_rank(tid + jj * FPS_BLOCK, bits);
                    if (d2 > bd || r < brank) { bd = d2; brank = r; }
                }
            }
            // keep the scheduler from hoisting every row's loads to the top
            if ((jj & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
        const unsigned long long key = (tid < nloc) ? (((unsigned long long)__float_as_uint(bd) << 32) | (unsigned)(~brank)) : 0ull;
        const unsigned long long win = block_max_key(key, slots, j & 1);
        last = fps_unrank(~(unsigned)(win & 0xffffffffu), bits);
        if (tid == 0) idx[j] = cl.n0 + last;
    }
    // write the running distances back (same side effect as :56).  Slots past the end hold -2; the
    // lane id is laundered so these addresses/masks are not kept alive across the sample loop.
    int tid_o = tid;
    asm volatile("" : "+v"(tid_o));
#pragma unroll
    for (int jj = 0; jj < PER; jj++)
        if (t[jj] >= 0.f) tmp[cl.n0 + tid_o + jj * FPS_BLOCK] = t[jj];
}

// any cloud size: running distances stay in tmp[] (L2-resident), as in the reference
__global__ __launch_bounds__(FPS_BLOCK) void fps_stream_kernel(int bits, const float* __restrict__ xyz,
                                                               const int* __restrict__ offset, const int* __restrict__ new_offset,
                                                               float* __restrict__ tmp, int* __restrict__ idx)
{
    __shared__ unsigned long long slots[2][FPS_WAVES];
    const FpsCloud cl = fps_cloud(offset, new_offset);
    if (cl.m1 <= cl.m0) return;
    const int tid = threadIdx.x, nloc = cl.n1 - cl.n0;
    const float* __restrict__ P = xyz + (size_t)3 * cl.n0;
    float* __restrict__ T = tmp + cl.n0;
    if (tid == 0) idx[cl.m0] = cl.n0;
    int last = 0;
    for (int j = cl.m0 + 1; j < cl.m1; j++) {
        const float lx = P[3 * last + 0], ly = P[3 * last + 1], lz = P[3 * last + 2];
        float bd = -1.f; unsigned brank = 0xffffffffu; bool any = false;
        for (int kk = tid; kk < nloc; kk += FPS_BLOCK) {
            const float d = cbl_dist2(P[3 * kk + 0], P[3 * kk + 1], P[3 * kk + 2], lx, ly, lz);
            const float d2 = fminf(d, T[kk]);
            T[kk] = d2;
            if (d2 >= bd) {
                const unsigned r = fps_rank(kk, bits);
                if (d2 > bd || r < brank) { bd = d2; brank = r; }
            }
            any = true;
        }
        const unsigned long long key = any ? (((unsigned long long)__float_as_uint(bd) << 32) | (unsigned)(~brank)) : 0ull;
        const unsigned long long win = block_max_key(key, slots, j & 1);
        last = fps_unrank(~(unsigned)(win & 0xffffffffu), bits);
        if (tid == 0) idx[j] = cl.n0 + last;
    }
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

CBL_EXPORT int cbl_furthestsampling(int b, int n_max, const float* xyz, const int* offset, const int* new_offset,
                                    float* tmp, int* idx, void* stream)
{
    if (b < 0 || n_max < 0) return CBL_ERR_BAD_ARG;
    if (b == 0) return CBL_OK;
    if (!xyz || !offset || !new_offset || !tmp || !idx) return CBL_ERR_BAD_ARG;
    const int B = ref_block_threads(n_max);
    int bits = 0; while ((1 << bits) < B) bits++;
    hipStream_t st = cbl_stream(stream);
    const dim3 grid(b), block(FPS_BLOCK);
    if (n_max <= 1 * FPS_BLOCK)       hipLaunchKernelGGL((fps_reg_kernel<1, true>),  grid, block, 0, st, bits, xyz, offset, new_offset, tmp, idx);
    else if (n_max <= 4 * FPS_BLOCK)  hipLaunchKernelGGL((fps_reg_kernel<4, true>),  grid, block, 0, st, bits, xyz, offset, new_offset, tmp, idx);
    else if (n_max <= 10 * FPS_BLOCK) hipLaunchKernelGGL((fps_reg_kernel<10, true>), grid, block, 0, st, bits, xyz, offset, new_offset, tmp, idx);
    else if (n_max <= 16 * FPS_BLOCK) hipLaunchKernelGGL((fps_reg_kernel<16, true>), grid, block, 0, st, bits, xyz, offset, new_offset, tmp, idx);
    else if (n_max <= 40 * FPS_BLOCK) hipLaunchKernelGGL((fps_reg_kernel<40, false>), grid, block, 0, st, bits, xyz, offset, new_offset, tmp, idx);
    else                              hipLaunchKernelGGL(fps_stream_kernel,          grid, block, 0, st, bits, xyz, offset, new_offset, tmp, idx);
    return cbl_status();
}
