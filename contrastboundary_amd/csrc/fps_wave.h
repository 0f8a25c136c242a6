// Cross-lane maxima / minima of the sampling kernels (fps.hip, fps_bucket.hip) on the VALU's data-parallel primitives, ONE instruction per step:
// xor-butterfly inside each 16-lane row (quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror, row_mirror), then row_bcast:15 / row_bcast:31 fold the four rows
// into lane 63.  A step written as max(v, dpp_mov(v)) with old = v compiles to v_mov + s_nop + v_mov_dpp + v_max (and for floats a canonicalising v_max on
// top: five issue slots); with old = 0 and bound_ctrl the move folds into the operation: v_max_i32_dpp v, v, v.  The sample loop is bound by its waves'
// vector issue slots (sixteen waves on four SIMDs), and these reductions were a third of them.
//   * running distances are non-negative floats (sums of squares; never -0) or negative sentinels (-2, -3: "no point"), so the signed-integer order of
//     the bit patterns is the float order wherever a valid value takes part, and a valid value always beats a sentinel: v_max_i32.
//   * an unsigned minimum is the complement of the maximum of the complements: v_max_u32, for which the 0 a lane without a source reads is neutral.
//   * row_bcast with every row enabled also writes rows 0 and 2 (garbage there); lane 63, the only lane read afterwards, sees exactly the values it
//     saw with the row masks.
#pragma once
#include <hip/hip_runtime.h>

namespace {

template <int CTRL> __device__ __forceinline__ int fw_max_i(int v) { return max(v, __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true)); }
template <int CTRL> __device__ __forceinline__ unsigned fw_max_u(unsigned v) { return max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true)); }

// every lane of a 16-lane row gets the row's maximum
__device__ __forceinline__ float row_max_f(float f)
{
    int v = __float_as_int(f);
    v = fw_max_i<0xB1>(v); v = fw_max_i<0x4E>(v); v = fw_max_i<0x141>(v); v = fw_max_i<0x140>(v);
    return __int_as_float(v);
}
__device__ __forceinline__ unsigned row_max_u(unsigned v) { v = fw_max_u<0xB1>(v); v = fw_max_u<0x4E>(v); v = fw_max_u<0x141>(v); v = fw_max_u<0x140>(v); return v; }
__device__ __forceinline__ unsigned row_min_u(unsigned v) { return ~row_max_u(~v); }

// wave-uniform results (read from lane 63)
__device__ __forceinline__ float wave_max_f(float f)
{
    int v = __float_as_int(row_max_f(f));
    v = fw_max_i<0x142>(v);                                          // rows 1, 3 <- lane 15 of the row below
    v = fw_max_i<0x143>(v);                                          // rows 2, 3 <- lane 31
    return __int_as_float(__builtin_amdgcn_readlane(v, 63));
}
__device__ __forceinline__ unsigned wave_max_u(unsigned v)
{
    v = row_max_u(v); v = fw_max_u<0x142>(v); v = fw_max_u<0x143>(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_min_u(unsigned v) { return ~wave_max_u(~v); }

}  // namespace
