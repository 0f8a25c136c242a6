// VALU issue rates on gfx950: v_fma_f32 vs v_pk_fma_f32 vs v_mov_dpp, one wave per SIMD and 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a)
{
    float x[16]; v2f y[8];
    for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 8; i++) y[i] = v2f{x[2 * i], x[2 * i + 1]};
    const v2f av = {a, a * 0.5f};
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = __builtin_fmaf(x[i], a, 1.0f);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; i++) y[i] = __builtin_elementwise_fma(y[i], av, av);
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[(i + 1) & 15]), 0x121, 0xf, 0xf, true));
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += x[i];
    for (int i = 0; i < 8; i++) s += y[i].x + y[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int blocks, float flop_per_it)
{
    float* out; hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0001f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = blocks * 4.0, instr = (double)iters * 16;       // wave-instructions per wave (mode 1: 8 packed)
    const double per_simd = waves / 1024.0;
    const double clk = ms * 1e-3 * 2.4e9;
    printf("%-14s blocks %5d (%.0f waves/SIMD): %.3f ms  %.2f TFLOP/s  ~%.2f clk per wave-instruction at 2.4 GHz\n", name, blocks, per_simd, ms,
           blocks * 256.0 * iters * flop_per_it / (ms * 1e-3) / 1e12, clk / ((MODE == 1 ? instr / 2 : instr) * per_simd));
    hipFree(out);
}
int main()
{
    for (int blocks : {256, 512, 1024, 2048}) {
        run<0>("v_fma_f32", blocks, 32.f);
        run<1>("v_pk_fma_f32", blocks, 32.f);
        run<2>("v_add+dpp", blocks, 16.f);
    }
    return 0;
}
