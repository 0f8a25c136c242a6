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
// The maximum, one instruction per step as well: a float's bits, with the magnitude bits flipped for negative values, order like signed integers (-0 below +0, no
// NaNs expected), and v_max_i32_dpp needs no canonicalising v_max in front of it (fmaxf(v, dpp_mov(v)) is five issue slots per step).  The 0 a lane without a
// source reads under row_bcast stands for +0.0 and only reaches lanes that are not read afterwards (see group_sum below).
template <int CTRL> __device__ __forceinline__ int dpp_max0_i(int v) { return max(v, __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true)); }
template <int G> __device__ __forceinline__ float group_max(float f)
{
    const int b = __float_as_int(f);
    int v = b ^ ((b >> 31) & 0x7fffffff);
    v = dpp_max0_i<0xB1>(v); v = dpp_max0_i<0x4E>(v); v = dpp_max0_i<0x141>(v); v = dpp_max0_i<0x140>(v);
    if (G >= 32) v = dpp_max0_i<0x142>(v);
    if (G == 32) v = __shfl(v, 31, 32);
    if (G == 64) { v = dpp_max0_i<0x143>(v); v = __builtin_amdgcn_readlane(v, 63); }
    return __int_as_float(v ^ ((v >> 31) & 0x7fffffff));
}

// The sum, one instruction per step: with old = 0 and bound_ctrl the move folds into the add (v_add_f32_dpp v, v, v; written as `v + dpp_mov(v)` with old = v
// it is v_mov + s_nop + v_mov_dpp + v_add, and the row steps a select on top).  Same butterfly, same operand order, same bits as group_reduce_f<G>(v, OpSumF()):
// with every row enabled row_bcast also adds into rows 0 and 2 (0 for a lane without a source, garbage in row 2), but the lanes read afterwards — 31 / 63 of
// each 32-lane half (G = 32), 63 (G = 64) — receive exactly the terms they received with the row masks.
template <int CTRL> __device__ __forceinline__ float dpp_add0(float v)
{
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int G> __device__ __forceinline__ float group_sum(float v)
{
    v = dpp_add0<0xB1>(v); v = dpp_add0<0x4E>(v); v = dpp_add0<0x141>(v); v = dpp_add0<0x140>(v);
    if (G == 16) return v;
    v = dpp_add0<0x142>(v);
    if (G == 32) return __shfl(v, 31, 32);
    v = dpp_add0<0x143>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
template <int G> __device__ __forceinline__ int group_sum_i(int v)
{
    // small non-negative counts: exact in fp32
    return (int)group_sum<G>((float)v);
}

}  // namespace
