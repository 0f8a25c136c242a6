// gfx9 instructions the grid searches (knn_grid.hip) name directly.  tests/host_emul/wave/knn_wave.h is the host stand-in with the same names.
#pragma once
#include <hip/hip_runtime.h>

// v_ffbl_b32: index of the lowest set bit of a 32-bit value, -1 when the value is 0
__device__ __forceinline__ int kw_ffbl(unsigned v) { int l; asm("v_ffbl_b32 %0, %1" : "=v"(l) : "v"(v)); return l; }
// a wave-uniform value placed in a vector register once: a select between a scalar and a vector under a lane mask needs two scalar operands (gfx9 allows
// one), so the compiler would copy the scalar at every use
__device__ __forceinline__ int kw_in_vgpr(int s) { int v; asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s)); return v; }
