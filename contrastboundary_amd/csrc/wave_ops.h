// Wave-level reductions on the VALU's data-parallel primitives (DPP), shared by the CBL kernels (gfx950, wave = 64).
#pragma once
#include <hip/hip_runtime.h>

namespace {

// Reductions over a G-lane group (G = 16, 32, 64 consecutive lanes) on the VALU's data-parallel primitives instead of LDS-crossbar
// shuffles: xor-butterfly inside each 16-lane row (quad_perm, row_half_mirror, row_mirror), then row_bcast:15 / :31 folds the rows;
// the group's last lane ends up with the total, which is handed to every lane of the group by v_readlane (G = 64) or one
// ds_bpermute (G = 32).
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int dpp_mov_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false); }
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_mov_f(float v) { return __int_as_float(dpp_mov_i<CTRL, ROW_MASK>(__float_as_int(v))); }

struct OpMaxF { __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); } };
struct OpSumF { __device__ __forceinline__ float operator()(float a, float b) const { return a + b; } };

template <int G, class Op> __device__ __forceinline__ float group_reduce_f(float v, Op op)
{
    v = op(v, dpp_mov_f<0xB1, 0xf>(v)); v = op(v, dpp_mov_f<0x4E, 0xf>(v)); v = op(v, dpp_mov_f<0x141, 0xf>(v)); v = op(v, dpp_mov_f<0x140, 0xf>(v));
    if (G == 16) return v;                                           // every lane of the row holds the row's result
    // row_bcast adds the previous row's lane 15 into a row: rows 1 and 3 first, then (G = 64) rows 2 and 3 take lane 31
    const float r1 = op(v, dpp_mov_f<0x142, 0xa>(v));
    v = ((threadIdx.x & 16) ? r1 : v);                               // DPP row_mask already restricts the write; keep rows 0, 2 unchanged
    if (G == 32) return __shfl(v, 31, 32);
    const float r2 = op(v, dpp_mov_f<0x143, 0xc>(v));
    v = ((threadIdx.x & 32) ? r2 : v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
template <int G> __device__ __forceinline__ float group_max(float v) { return group_reduce_f<G>(v, OpMaxF()); }
template <int G> __device__ __forceinline__ float group_sum(float v) { return group_reduce_f<G>(v, OpSumF()); }
template <int G> __device__ __forceinline__ int group_sum_i(int v)
{
    // small non-negative counts: exact in fp32
    return (int)group_reduce_f<G>((float)v, OpSumF());
}

}  // namespace
