// TEST INFRASTRUCTURE: host stand-in for contrastboundary_amd/csrc/gather_wave.h (same names, emulated on the fibre waves of hip/hip_runtime.h).
#pragma once
#include <hip/hip_runtime.h>

static inline int gw_shfl(int v, int src_lane)
{
    // every lane deposits (value, wanted lane) and takes the value of the lane it asked for
    const float pay[2] = {__int_as_float(v), __int_as_float(src_lane)};
    float r;
    emul::wave_collective(pay, 2, &r, 1, [](emul::Wave& w) { for (int l = 0; l < 64; l++) w.out[l][0] = w.in[__float_as_int(w.in[l][1]) & 63][0]; });
    return __float_as_int(r);
}
static inline int gw_uniform(int v)
{
    const float pay = __int_as_float(v);
    float r;
    emul::wave_collective(&pay, 1, &r, 1, [](emul::Wave& w) { int first = 0; while (first < 63 && !w.present[first]) first++; for (int l = 0; l < 64; l++) w.out[l][0] = w.in[first][0]; });
    return __float_as_int(r);
}
static inline int gw_readlane(int v, int src_lane) { return gw_shfl(v, src_lane); }      // every active lane asks for the same lane
static inline void gw_wave_sync() { float z = 0.f, r; emul::wave_collective(&z, 1, &r, 1, [](emul::Wave&) {}); }
static inline void gw_store16_streaming(float* dst, const float* src) { std::memcpy(dst, src, 16); }
