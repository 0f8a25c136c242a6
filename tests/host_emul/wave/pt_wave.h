// TEST INFRASTRUCTURE: host stand-in for contrastboundary_amd/csrc/pt_wave.h (same names, emulated on the fibre waves of hip/hip_runtime.h).
#pragma once
#include <hip/hip_runtime.h>

struct pt_f32x4 { float v[4]; float& operator[](int i) { return v[i]; } const float& operator[](int i) const { return v[i]; } };
static inline pt_f32x4 pt_vec4(float a, float b, float c, float d) { pt_f32x4 r; r.v[0] = a; r.v[1] = b; r.v[2] = c; r.v[3] = d; return r; }

static inline pt_f32x4 pt_mfma(float a, float b, pt_f32x4 c)
{
    const float pay[6] = {a, b, c[0], c[1], c[2], c[3]};
    pt_f32x4 d;
    emul::wave_collective(pay, 6, d.v, 4, [](emul::Wave& w) {
        for (int l = 0; l < 64; l++)
            for (int v = 0; v < 4; v++) {
                const int row = 4 * (l / 16) + v, col = l % 16;
                float acc = w.in[l][2 + v];
                for (int k = 0; k < 4; k++) acc = std::fmaf(w.in[row + 16 * k][0], w.in[16 * k + col][1], acc);   // A[row][k] lane row + 16k, B[k][col] lane 16k + col
                w.out[l][v] = acc;
            }
    });
    return d;
}

template <class M> static inline float pt_lane_move(float v, M src_of)
{
    float r;
    emul::wave_collective(&v, 1, &r, 1, [&](emul::Wave& w) { for (int l = 0; l < 64; l++) w.out[l][0] = w.in[src_of(l)][0]; });
    return r;
}
static inline float pt_quad_xor1(float v) { return pt_lane_move(v, [](int l) { return l ^ 1; }); }
static inline float pt_quad_xor2(float v) { return pt_lane_move(v, [](int l) { return l ^ 2; }); }
static inline float pt_half_mirror(float v) { return pt_lane_move(v, [](int l) { return (l & ~7) | (7 - (l & 7)); }); }
static inline float pt_row_mirror(float v) { return pt_lane_move(v, [](int l) { return (l & ~15) | (15 - (l & 15)); }); }
static inline float pt_row_ror4(float v) { return pt_lane_move(v, [](int l) { return (l & ~15) | ((l - 4) & 15); }); }
static inline float pt_row_ror8(float v) { return pt_lane_move(v, [](int l) { return (l & ~15) | ((l - 8) & 15); }); }
static inline float pt_xor16(float v) { return pt_lane_move(v, [](int l) { return l ^ 16; }); }
static inline float pt_xor32(float v) { return pt_lane_move(v, [](int l) { return l ^ 32; }); }
static inline void pt_wave_sync() { float z = 0.f, r; emul::wave_collective(&z, 1, &r, 1, [](emul::Wave&) {}); }
