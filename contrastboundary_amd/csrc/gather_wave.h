// cross-lane / memory primitives of query_group_pipe.h and k4_rows_pipe.h (device build).  tests/host_emul/wave/gather_wave.h is the host stand-in with the same names.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ int gw_shfl(int v, int src_lane) { return __shfl(v, src_lane); }
__device__ __forceinline__ int gw_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// the value `v` holds in lane `src_lane` (wave-uniform lane index)
__device__ __forceinline__ int gw_readlane(int v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }
__device__ __forceinline__ void gw_wave_sync() { __builtin_amdgcn_wave_barrier(); }
// 16 aligned bytes (four floats) from `src` (LDS) to `dst` (global) as a streaming store
__device__ __forceinline__ void gw_store16_streaming(float* dst, const float* src)
{
    typedef float v4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(*reinterpret_cast<const v4*>(src), reinterpret_cast<v4*>(dst));
}
