// TEST INFRASTRUCTURE (not product code): the cross-lane builtins the sampling kernels (contrastboundary_amd/csrc/fps_bucket.hip, fps_wave.h) use directly,
// emulated on the fibre waves of hip/hip_runtime.h — every call is a rendezvous of the wave's alive lanes.  Semantics follow the gfx9 ISA:
//   update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl): lane l of an enabled row reads src of lane s(l); a lane without a source keeps `old`, or gets 0
//   with bound_ctrl; lanes of rows the row_mask disables keep `old` (bank_mask is always 0xf in our kernels).  Controls: quad_perm (0x00..0xFF),
//   row_shl / row_shr / row_ror:n (0x101.. / 0x111.. / 0x121..), wave_shl / rol / shr / ror:1 (0x130 / 0x134 / 0x138 / 0x13C), row_mirror (0x140),
//   row_half_mirror (0x141), row_bcast:15 (0x142), row_bcast:31 (0x143).
#pragma once
#define __HIPCC__ 1                                                  // our headers that also serve plain-C++ builds (grid_core.h) take their device branch
#include <hip/hip_runtime.h>

static inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline unsigned __brev(unsigned v) { unsigned r = 0; for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i); return r; }
#define __builtin_amdgcn_sched_barrier(x) ((void)0)

namespace emul {
// one int per lane in, one int per lane out, computed from all lanes' inputs
// tag / scope: which collective this is and whether it spans the wave (1) or a lane group (0) — see hip_runtime.h resolve()
template <class F> inline int lanes_i(int v, int aux, F f, unsigned tag = 0, int scope = 1)
{
    float pay[2] = {__int_as_float(v), __int_as_float(aux)}, r;
    wave_collective(pay, 2, &r, 1, [&](Wave& w) {
        int in[WAVE], ax[WAVE], out[WAVE];
        for (int l = 0; l < WAVE; l++) { in[l] = __float_as_int(w.in[l][0]); ax[l] = __float_as_int(w.in[l][1]); }
        f(in, ax, out, w.present);
        for (int l = 0; l < WAVE; l++) w.out[l][0] = __int_as_float(out[l]);
    }, tag, scope);
    return __float_as_int(r);
}
inline int dpp_source(int l, int ctrl)                              // -1: no source
{
    const int row = l & ~15, i = l & 15;
    if (ctrl >= 0 && ctrl <= 0xFF) return (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
    if (ctrl >= 0x101 && ctrl <= 0x10F) return (i + (ctrl & 15) <= 15) ? l + (ctrl & 15) : -1;      // row_shl:n — lane i reads lane i + n of its row
    if (ctrl >= 0x111 && ctrl <= 0x11F) return (i - (ctrl & 15) >= 0) ? l - (ctrl & 15) : -1;       // row_shr:n — lane i reads lane i - n of its row
    if (ctrl >= 0x121 && ctrl <= 0x12F) return row | ((i - (ctrl & 15)) & 15);      // row_ror:n — lane i reads lane i - n (mod 16)
    if (ctrl == 0x130) return l < 63 ? l + 1 : -1;                                   // wave_shl:1
    if (ctrl == 0x134) return (l + 1) & 63;                                          // wave_rol:1
    if (ctrl == 0x138) return l > 0 ? l - 1 : -1;                                    // wave_shr:1
    if (ctrl == 0x13C) return (l - 1) & 63;                                          // wave_ror:1
    if (ctrl == 0x140) return row | (15 - i);
    if (ctrl == 0x141) return (l & ~7) | (7 - (l & 7));
    if (ctrl == 0x142) return (l >= 16) ? row - 1 : -1;                              // lane 15 of the row below
    if (ctrl == 0x143) return (l >= 32) ? 31 : -1;
    std::fprintf(stderr, "emul: DPP control 0x%x not emulated\n", ctrl); std::abort();
}
}  // namespace emul

static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
    (void)bank_mask;
    return emul::lanes_i(src, old, [&](const int* in, const int* ax, int* out, const bool* present) {
        for (int l = 0; l < emul::WAVE; l++) {
            if (!((row_mask >> (l >> 4)) & 1)) { out[l] = ax[l]; continue; }
            const int s = emul::dpp_source(l, ctrl);
            out[l] = (s >= 0 && present[s]) ? in[s] : (bound_ctrl ? 0 : ax[l]);
        }
    }, 0x10000u | (unsigned)ctrl, (ctrl == 0x130 || ctrl == 0x134 || ctrl == 0x138 || ctrl == 0x13C) ? 1 : 0);
}
static inline int __builtin_amdgcn_readlane(int v, int lane)
{
    return emul::lanes_i(v, lane, [&](const int* in, const int* ax, int* out, const bool* present) {
        for (int l = 0; l < emul::WAVE; l++) out[l] = present[l] ? in[ax[l] & 63] : 0;
    });
}
static inline int __builtin_amdgcn_readfirstlane(int v)
{
    return emul::lanes_i(v, 0, [&](const int* in, const int*, int* out, const bool* present) {
        int first = 0; while (first < emul::WAVE - 1 && !present[first]) first++;
        for (int l = 0; l < emul::WAVE; l++) out[l] = in[first];
    });
}
static inline unsigned long long __ballot(int pred)
{
    const int lo = emul::lanes_i(pred ? 1 : 0, 0, [&](const int* in, const int*, int* out, const bool* present) {
        unsigned m = 0; for (int l = 0; l < 32; l++) if (present[l] && in[l]) m |= 1u << l;
        for (int l = 0; l < emul::WAVE; l++) out[l] = (int)m;
    });
    const int hi = emul::lanes_i(pred ? 1 : 0, 0, [&](const int* in, const int*, int* out, const bool* present) {
        unsigned m = 0; for (int l = 32; l < 64; l++) if (present[l] && in[l]) m |= 1u << (l - 32);
        for (int l = 0; l < emul::WAVE; l++) out[l] = (int)m;
    });
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}

// ---- wave shuffles (ds_bpermute-backed on the device): width = 64 or a power of two below it; lanes of a width-group exchange among themselves
namespace emul {
template <class T, class F> inline T lanes_any(T v, F src_of, int width = 64)
{
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "4- or 8-byte values");
    float pay[2] = {0.f, 0.f}, r[2];
    std::memcpy(pay, &v, sizeof(T));
    wave_collective(pay, 2, r, 2, [&](Wave& w) {
        for (int l = 0; l < WAVE; l++) { const int s = src_of(l); const int t = (s >= 0 && s < WAVE && w.present[s]) ? s : l; w.out[l][0] = w.in[t][0]; w.out[l][1] = w.in[t][1]; }
    }, 0x20000u | (unsigned)width, width < 64 ? 0 : 1);
    T o; std::memcpy(&o, r, sizeof(T));
    return o;
}
}  // namespace emul
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) { return emul::lanes_any(v, [&](int l) { return l ^ mask; }, width); }
template <class T> static inline T __shfl(T v, int src, int width = 64)
{
    // every lane names its own source: the lane index travels with the value
    float pay[3] = {0.f, 0.f, __int_as_float(src)}, r[2];
    std::memcpy(pay, &v, sizeof(T));
    emul::wave_collective(pay, 3, r, 2, [&](emul::Wave& w) {
        for (int l = 0; l < emul::WAVE; l++) {
            const int s = (l & ~(width - 1)) | (__float_as_int(w.in[l][2]) & (width - 1));
            const int t = w.present[s] ? s : l;
            w.out[l][0] = w.in[t][0]; w.out[l][1] = w.in[t][1];
        }
    }, 0x30000u | (unsigned)width, width < 64 ? 0 : 1);
    T o; std::memcpy(&o, r, sizeof(T));
    return o;
}
static inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / std::sqrt(x); }

// ---- v_mfma_f32_16x16x4_f32 on a wave of fibres: D = A (16 x 4) . B (4 x 16) + C, operands and results in the instruction's own lane layout
//   A[i][k] in lane i + 16 k (one float), B[k][j] in lane 16 k + j (one float), C / D[4 (lane / 16) + r][lane % 16] in element r of the lane's four
typedef float emul_f32x4 __attribute__((vector_size(16)));
static inline emul_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, emul_f32x4 c, int, int, int)
{
    const float pay[6] = {a, b, c[0], c[1], c[2], c[3]};
    float d[4];
    emul::wave_collective(pay, 6, d, 4, [](emul::Wave& w) {
        for (int l = 0; l < 64; l++)
            for (int v = 0; v < 4; v++) {
                const int row = 4 * (l / 16) + v, col = l % 16;
                float acc = w.in[l][2 + v];
                for (int k = 0; k < 4; k++) acc = std::fmaf(w.in[row + 16 * k][0], w.in[16 * k + col][1], acc);
                w.out[l][v] = acc;
            }
    });
    return emul_f32x4{d[0], d[1], d[2], d[3]};
}
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }

// ---- atomics (fibres are cooperative: a read-modify-write is never interleaved), lane-position counts, the LDS crossbar's push form, fast math
template <class T> static inline T atomicAdd(T* p, T v) { const T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { const T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { const T o = *p; if (v < o) *p = v; return o; }
static inline void unsafeAtomicAdd(float* p, float v) { *p += v; }
static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned base)
{
    const int lane = emul::S().cur->lane;
    return base + (unsigned)__builtin_popcount(lane >= 32 ? mask : (mask & ((1u << lane) - 1u)));
}
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned base)
{
    const int lane = emul::S().cur->lane;
    return base + (lane > 32 ? (unsigned)__builtin_popcount(mask & ((1u << (lane - 32)) - 1u)) : 0u);
}
// ds_permute_b32 (forward / push): lane l sends `v` to lane (addr / 4) % 64; a lane nobody sends to reads 0; of several senders the highest lane wins
static inline int __builtin_amdgcn_ds_permute(int addr, int v)
{
    return emul::lanes_i(v, addr, [&](const int* in, const int* ax, int* out, const bool* present) {
        for (int l = 0; l < emul::WAVE; l++) out[l] = 0;
        for (int l = 0; l < emul::WAVE; l++) if (present[l]) out[(ax[l] >> 2) & 63] = in[l];
    });
}
static inline int __builtin_amdgcn_ds_bpermute(int addr, int v)
{
    return emul::lanes_i(v, addr, [&](const int* in, const int* ax, int* out, const bool* present) {
        for (int l = 0; l < emul::WAVE; l++) { const int s = (ax[l] >> 2) & 63; out[l] = present[s] ? in[s] : 0; }
    });
}
static inline float __builtin_amdgcn_sqrtf(float x) { return std::sqrt(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_exp2f(float x) { return std::exp2(x); }

// ---- what local_aggregation.hip / kpconv_backward.hip use beyond the above
struct float3 { float x, y, z; };
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline unsigned min(unsigned a, int b) { return b < 0 ? 0u : (a < (unsigned)b ? a : (unsigned)b); }
static inline unsigned min(int a, unsigned b) { return min(b, a); }
static inline long long min(long long a, int b) { return a < b ? a : (long long)b; }
static inline long long min(int a, long long b) { return a < b ? (long long)a : b; }
static inline long long max(long long a, int b) { return a > b ? a : (long long)b; }
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
static inline void __builtin_amdgcn_wave_barrier() { float z = 0.f, r; emul::wave_collective(&z, 1, &r, 1, [](emul::Wave&) {}); }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
// v_permlane16_swap / v_permlane32_swap: rows 1, 3 (lanes 32..63) of the first operand change places with rows 0, 2 (lanes 0..31) of the second; {first', second'}
struct emul_u32x2 { unsigned v[2]; unsigned operator[](int i) const { return v[i]; } };
static inline emul_u32x2 emul_permlane_swap(unsigned a, unsigned b, int span)
{
    emul_u32x2 o;
    float pay[2] = {__int_as_float((int)a), __int_as_float((int)b)}, r[2];
    emul::wave_collective(pay, 2, r, 2, [&](emul::Wave& w) {
        for (int l = 0; l < emul::WAVE; l++) {
            const bool upper = (l / span) & 1;                      // the half of a 2 * span group that is exchanged
            // first' : lower part keeps first, upper part takes second's lower part;  second' : lower part takes first's upper part, upper part keeps second
            w.out[l][0] = upper ? w.in[l - span][1] : w.in[l][0];
            w.out[l][1] = upper ? w.in[l][1] : w.in[l + span][0];
        }
    });
    o.v[0] = (unsigned)__float_as_int(r[0]); o.v[1] = (unsigned)__float_as_int(r[1]);
    return o;
}
static inline emul_u32x2 __builtin_amdgcn_permlane16_swap(unsigned a, unsigned b, bool, bool) { return emul_permlane_swap(a, b, 16); }
static inline emul_u32x2 __builtin_amdgcn_permlane32_swap(unsigned a, unsigned b, bool, bool) { return emul_permlane_swap(a, b, 32); }
// v_sin_f32 takes its argument in REVOLUTIONS; v_fract_f32
static inline float __builtin_amdgcn_sinf(float x) { return (float)std::sin(6.283185307179586476925 * (double)x); }
static inline float __builtin_amdgcn_cosf(float x) { return (float)std::cos(6.283185307179586476925 * (double)x); }
static inline float __builtin_amdgcn_fractf(float x) { return x - std::floor(x); }

// ---- the rest of what the search / table / gather files use
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int __any(int pred) { return __ballot(pred) != 0ull; }
static inline int __all(int pred)
{
    return emul::lanes_i(pred ? 1 : 0, 0, [&](const int* in, const int*, int* out, const bool* present) {
        int a = 1; for (int l = 0; l < emul::WAVE; l++) if (present[l] && !in[l]) a = 0;
        for (int l = 0; l < emul::WAVE; l++) out[l] = a;
    });
}
#define __hip_atomic_load(ptr, order, scope) (*(ptr))
#define __hip_atomic_store(ptr, v, order, scope) (*(ptr) = (v))
static inline void __threadfence_block() {}
static inline void __threadfence() {}
static inline long long __double_as_longlong(double d) { long long v; std::memcpy(&v, &d, 8); return v; }
static inline double __longlong_as_double(long long v) { double d; std::memcpy(&d, &v, 8); return d; }
// (byte copies: a template argument loses the `aligned(4)` of the kernels' unaligned 16-byte vector type, and g++ would store it with movaps)
template <class T> static inline void __builtin_nontemporal_store(T v, T* p) { std::memcpy((void*)p, &v, sizeof(T)); }
template <class T> static inline T __builtin_nontemporal_load(const T* p) { T v; std::memcpy(&v, (const void*)p, sizeof(T)); return v; }
template <class T> static inline T __shfl_up(T v, unsigned delta, int width = 64)
{
    return emul::lanes_any(v, [&](int l) { const int i = l & (width - 1); return i >= (int)delta ? l - (int)delta : l; }, width);
}
template <class T> static inline T __shfl_down(T v, unsigned delta, int width = 64)
{
    return emul::lanes_any(v, [&](int l) { const int i = l & (width - 1); return i + (int)delta < width ? l + (int)delta : l; }, width);
}
// ds_swizzle_b32, bit-mask mode (offset bit 15 = 0): inside every group of 32 lanes, lane' = ((lane & and_mask) | or_mask) ^ xor_mask
static inline int __builtin_amdgcn_ds_swizzle(int v, int pattern)
{
    if (pattern & 0x8000) { std::fprintf(stderr, "emul: ds_swizzle quad-perm mode not emulated\n"); std::abort(); }
    const int and_mask = pattern & 31, or_mask = (pattern >> 5) & 31, xor_mask = (pattern >> 10) & 31;
    return emul::lanes_i(v, 0, [&](const int* in, const int*, int* out, const bool* present) {
        for (int l = 0; l < emul::WAVE; l++) { const int s = (l & 32) | ((((l & 31) & and_mask) | or_mask) ^ xor_mask); out[l] = present[s] ? in[s] : 0; }
    });
}
typedef void* hipEvent_t;
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
constexpr unsigned hipEventDisableTiming = 2;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
static inline hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(dst, src, n); return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
template <class F> static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
template <class T> static inline T atomicSub(T* p, T v) { const T o = *p; *p = o - v; return o; }
template <class T> static inline T atomicOr(T* p, T v) { const T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicExch(T* p, T v) { const T o = *p; *p = v; return o; }
template <class T> static inline T atomicCAS(T* p, T cmp, T v) { const T o = *p; if (o == cmp) *p = v; return o; }
