// a4, dense part: the Linear layers INSIDE the vector attention act on (n*K) rows with tiny feature widths —
//   linear_p: Linear(3,3), Linear(3,C)      linear_w: Linear(C, C/8), Linear(C/8, C/8)      /root/reference/pytorch/model/blocks.py:23-28,38-40
// i.e. GEMMs of shape (163 840 x 3) @ (3 x 3) or a weight gradient (8 x 163 840) @ (163 840 x 64).  A tiled GEMM library has no good
// kernel for that (rocBLAS: 0.4-0.5 ms per call at n*K = 163 840, profiles/): they are streaming problems — read the rows once, do a
// few FMAs per element — so they are written as such:
//   forward / input gradient:  one lane per output element, weights in LDS (row stride padded), the row's inputs read through L1
//                              (the C_out lanes of a row read the same addresses);
//   weight / bias gradient:    a workgroup stages tiles of rows of x and dy in LDS, every lane owns a few (c_out, c_in) pairs and
//                              accumulates over the workgroup's rows in registers; the <= 768 workgroup partials are summed in fp64
//                              by a second small kernel (plain stores: no pre-zeroed outputs, no atomics).
// fp32 FMA chains in index order; results differ from a GEMM library's by summation order only.
#include "cbl_common.h"

namespace {

constexpr int SL_BLOCK = 256;
constexpr int SL_MAX_W = 4096;                 // c_in * c_out handled here (weights + padding must fit LDS comfortably)

// y[r, c] = sum_a x[r, a] * W(c, a) (+ b[c]);   W(c, a) = w[c * cin + a]  (TRANS = false: y = x W^T, nn.Linear)
//                                                        w[a * cout + c]  (TRANS = true:  y = x W,   the input gradient dy -> dx)
template <bool TRANS, bool VEC>                  // VEC: c_in % 4 == 0 and 16-byte aligned rows -> float4 reads of the row
__global__ __launch_bounds__(SL_BLOCK) void skinny_linear_kernel(long long rows, int cin, int cout, const float* __restrict__ x,
                                                                 const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ y)
{
    extern __shared__ float ws[];                                   // [cout][cin + 1]
    const int ld = cin + 1;
    for (int e = threadIdx.x; e < cin * cout; e += SL_BLOCK) {
        const int c = TRANS ? e % cout : e / cin, a = TRANS ? e / cout : e % cin;        // coalesced read of w
        ws[c * ld + a] = w[e];
    }
    __syncthreads();
    const long long total = rows * cout;
    for (long long e = (long long)blockIdx.x * SL_BLOCK + threadIdx.x; e < total; e += (long long)gridDim.x * SL_BLOCK) {
        const long long r = e / cout; const int c = (int)(e - r * cout);
        const float* __restrict__ xr = x + r * cin;
        const float* wr = ws + c * ld;
        float acc = b ? b[c] : 0.f;
        if (VEC) {
            for (int a = 0; a < cin; a += 4) {
                const float4 v = *reinterpret_cast<const float4*>(xr + a);
                acc += v.x * wr[a]; acc += v.y * wr[a + 1]; acc += v.z * wr[a + 2]; acc += v.w * wr[a + 3];
            }
        } else {
            for (int a = 0; a < cin; a++) acc += xr[a] * wr[a];
        }
        y[e] = acc;
    }
}

// partial[b][c*cin + a] = sum over workgroup b's rows of dy[r, c] * x[r, a];   partial[b][cin*cout + c] = sum of dy[r, c]
constexpr int SL_TILE = 64;                                         // rows per LDS tile
constexpr int SL_WGRAD_BLOCKS = 768;
__global__ __launch_bounds__(SL_BLOCK) void skinny_linear_wgrad_kernel(long long rows, int cin, int cout, const float* __restrict__ x,
                                                                       const float* __restrict__ dy, float* __restrict__ partial, int want_bias)
{
    extern __shared__ float tile[];                                 // x tile [SL_TILE][cin + 1], dy tile [SL_TILE][cout + 1]
    const int lx = cin + 1, ly = cout + 1;
    float* xs = tile; float* ys = tile + SL_TILE * lx;
    constexpr int PAIRS = SL_MAX_W / SL_BLOCK;                      // (c, a) pairs per lane
    const int npairs = cin * cout;
    float acc[PAIRS], accb = 0.f;
#pragma unroll
    for (int j = 0; j < PAIRS; j++) acc[j] = 0.f;
    const long long ntiles = (rows + SL_TILE - 1) / SL_TILE;
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long long r0 = t * SL_TILE;
        const int nr = (int)min((long long)SL_TILE, rows - r0);
        __syncthreads();                                            // previous tile fully consumed
        for (int e = threadIdx.x; e < nr * cin; e += SL_BLOCK) xs[(e / cin) * lx + e % cin] = x[r0 * cin + e];
        for (int e = threadIdx.x; e < nr * cout; e += SL_BLOCK) ys[(e / cout) * ly + e % cout] = dy[r0 * cout + e];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PAIRS; j++) {
            const int pidx = threadIdx.x + j * SL_BLOCK;
            if (pidx < npairs) {
                const int c = pidx / cin, a = pidx - c * cin;
                float s = 0.f;
                for (int r = 0; r < nr; r++) s += ys[r * ly + c] * xs[r * lx + a];
                acc[j] += s;
            }
        }
        if (want_bias && threadIdx.x < cout) {
            float s = 0.f;
            for (int r = 0; r < nr; r++) s += ys[r * ly + threadIdx.x];
            accb += s;
        }
    }
    float* mine = partial + (size_t)blockIdx.x * (npairs + cout);
#pragma unroll
    for (int j = 0; j < PAIRS; j++) {
        const int pidx = threadIdx.x + j * SL_BLOCK;
        if (pidx < npairs) mine[pidx] = acc[j];
    }
    if (threadIdx.x < cout) mine[npairs + threadIdx.x] = accb;
}

// dw / db = sum of the workgroup partials (fp64), 16 outputs x 16 slices per workgroup; plain stores: no pre-zeroed outputs, no atomics
__global__ __launch_bounds__(256) void skinny_linear_wgrad_finalize_kernel(int npairs, int cout, int nblocks, const float* __restrict__ partial,
                                                                           float* __restrict__ dw, float* __restrict__ db)
{
    __shared__ double red[16][16];
    const int e = blockIdx.x * 16 + (threadIdx.x & 15), js = threadIdx.x >> 4, total = npairs + cout;
    double a = 0.0;
    if (e < total) {
#pragma unroll 8
        for (int b = js; b < nblocks; b += 16) a += (double)partial[(size_t)b * total + e];
    }
    red[js][threadIdx.x & 15] = a;
    __syncthreads();
    if (js == 0 && e < total) {
        double s0 = 0.0;
        for (int j = 0; j < 16; j++) s0 += red[j][threadIdx.x & 15];
        if (e < npairs) dw[e] = (float)s0;
        else if (db) db[e - npairs] = (float)s0;
    }
}

int sl_check(long long rows, int cin, int cout)
{
    if (rows < 0 || cin <= 0 || cout <= 0) return CBL_ERR_BAD_ARG;
    if ((long long)cin * cout > SL_MAX_W || cin + cout > 200) return CBL_ERR_UNSUPPORTED;     // LDS tiles: 64 rows x (c_in + c_out + 2) floats
    return CBL_OK;
}

}  // namespace

CBL_EXPORT int cbl_skinny_linear_forward(long long rows, int cin, int cout, const float* x, const float* weight, const float* bias, float* y, void* stream)
{
    const int rc = sl_check(rows, cin, cout);
    if (rc) return rc;
    if (rows == 0) return CBL_OK;
    if (!x || !weight || !y) return CBL_ERR_BAD_ARG;
    const dim3 grid(cbl_grid_for(rows * cout, SL_BLOCK, 2048));
    const size_t lds = sizeof(float) * (size_t)cout * (cin + 1);
    if (cin % 4 == 0 && cbl_host_aligned16(x))
        hipLaunchKernelGGL((skinny_linear_kernel<false, true>), grid, dim3(SL_BLOCK), lds, cbl_stream(stream), rows, cin, cout, x, weight, bias, y);
    else
        hipLaunchKernelGGL((skinny_linear_kernel<false, false>), grid, dim3(SL_BLOCK), lds, cbl_stream(stream), rows, cin, cout, x, weight, bias, y);
    return cbl_status();
}

CBL_EXPORT int cbl_skinny_linear_backward_input(long long rows, int cin, int cout, const float* grad_y, const float* weight, float* grad_x, void* stream)
{
    const int rc = sl_check(rows, cin, cout);
    if (rc) return rc;
    if (rows == 0) return CBL_OK;
    if (!grad_y || !weight || !grad_x) return CBL_ERR_BAD_ARG;
    // dx = dy @ W: the same kernel with the roles of c_in / c_out swapped and W read transposed
    const dim3 grid(cbl_grid_for(rows * cin, SL_BLOCK, 2048));
    const size_t lds = sizeof(float) * (size_t)cin * (cout + 1);
    if (cout % 4 == 0 && cbl_host_aligned16(grad_y))
        hipLaunchKernelGGL((skinny_linear_kernel<true, true>), grid, dim3(SL_BLOCK), lds, cbl_stream(stream), rows, cout, cin, grad_y, weight, nullptr, grad_x);
    else
        hipLaunchKernelGGL((skinny_linear_kernel<true, false>), grid, dim3(SL_BLOCK), lds, cbl_stream(stream), rows, cout, cin, grad_y, weight, nullptr, grad_x);
    return cbl_status();
}

CBL_EXPORT size_t cbl_skinny_linear_workspace_bytes(int cin, int cout)
{
    if (sl_check(0, cin, cout) != CBL_OK) return 0;
    return sizeof(float) * (size_t)SL_WGRAD_BLOCKS * ((size_t)cin * cout + cout) + 256;
}

CBL_EXPORT int cbl_skinny_linear_backward_weight(long long rows, int cin, int cout, const float* x, const float* grad_y, float* grad_weight, float* grad_bias,
                                                 void* workspace, size_t workspace_bytes, void* stream)
{
    const int rc = sl_check(rows, cin, cout);
    if (rc) return rc;
    if (!grad_weight || !workspace) return CBL_ERR_BAD_ARG;
    if (workspace_bytes < cbl_skinny_linear_workspace_bytes(cin, cout)) return CBL_ERR_WORKSPACE;
    hipStream_t st = cbl_stream(stream);
    if (rows == 0) {
        (void)hipMemsetAsync(grad_weight, 0, sizeof(float) * (size_t)cin * cout, st);
        if (grad_bias) (void)hipMemsetAsync(grad_bias, 0, sizeof(float) * cout, st);
        return cbl_status();
    }
    if (!x || !grad_y) return CBL_ERR_BAD_ARG;
    const long long ntiles = (rows + SL_TILE - 1) / SL_TILE;
    const int nblocks = (int)min(ntiles, (long long)SL_WGRAD_BLOCKS);
    float* partial = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(skinny_linear_wgrad_kernel, dim3(nblocks), dim3(SL_BLOCK), sizeof(float) * (size_t)SL_TILE * (cin + 1 + cout + 1), st,
                       rows, cin, cout, x, grad_y, partial, grad_bias ? 1 : 0);
    hipLaunchKernelGGL(skinny_linear_wgrad_finalize_kernel, dim3(cbl_div_up(cin * cout + cout, 16)), dim3(256), 0, st, cin * cout, cout, nblocks, partial,
                       grad_weight, grad_bias);
    return cbl_status();
}
