// Cross-lane primitives of the Point-Transformer layer kernels (pt_layer.hip), gfx950, wave = 64: the f32 matrix instruction and the
// DPP / bpermute moves its tile layouts need.  One name per hardware operation, so that the kernels read as the data movement they do
// (and so that tests/host_emul/wave/pt_wave.h can stand in for this header when the kernels are compiled for the CPU test).
#pragma once
#include <hip/hip_runtime.h>

using pt_f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ pt_f32x4 pt_vec4(float a, float b, float c, float d) { pt_f32x4 r = {a, b, c, d}; return r; }

// v_mfma_f32_16x16x4_f32:  D = A (16 x 4) . B (4 x 16) + C.   lane l holds A[l % 16][l / 16], B[l / 16][l % 16], and D[4 (l / 16) + v][l % 16], v = 0..3.
// Bit for bit a k-ordered chain of fused multiply-adds: D = fma(A[.][3], B[3][.], fma(A[.][2], B[2][.], fma(A[.][1], B[1][.], fma(A[.][0], B[0][.], C)))).
__device__ __forceinline__ pt_f32x4 pt_mfma(float a, float b, pt_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int CTRL> __device__ __forceinline__ float pt_dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));     // old = 0 + bound_ctrl: every source lane of these permutations exists, and in this form the move folds into its user (v_add_f32_dpp)
}
// value of another lane of the same 16-lane row
__device__ __forceinline__ float pt_quad_xor1(float v) { return pt_dpp<0xB1>(v); }        // lane ^ 1   (quad_perm 1,0,3,2)
__device__ __forceinline__ float pt_quad_xor2(float v) { return pt_dpp<0x4E>(v); }        // lane ^ 2   (quad_perm 2,3,0,1)
__device__ __forceinline__ float pt_half_mirror(float v) { return pt_dpp<0x141>(v); }     // 7 - lane within each half row
__device__ __forceinline__ float pt_row_mirror(float v) { return pt_dpp<0x140>(v); }      // 15 - lane within the row
__device__ __forceinline__ float pt_row_ror4(float v) { return pt_dpp<0x124>(v); }        // row rotated by 4
__device__ __forceinline__ float pt_row_ror8(float v) { return pt_dpp<0x128>(v); }        // row rotated by 8
// value of the lane 16 / 32 away (another row of the wave)
__device__ __forceinline__ float pt_xor16(float v) { return __shfl_xor(v, 16, 64); }
__device__ __forceinline__ float pt_xor32(float v) { return __shfl_xor(v, 32, 64); }
// A point all lanes of the wave pass together.  LDS traffic of one wave is processed in program order and the compiler keeps may-alias LDS
// accesses in order, so on the device this emits nothing (a memory fence here would drain the wave's prefetched global loads: measured 2.5x);
// it marks the hand-over between the lanes that write a staged tile and the lanes that read it — the host stand-in of this header needs the rendezvous.
__device__ __forceinline__ void pt_wave_sync() { __builtin_amdgcn_wave_barrier(); }
