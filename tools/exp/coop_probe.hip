// probe: cost of a cooperative launch with 4 grid-wide barriers vs 5 plain launches (decides whether the grid build can be one kernel)
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <chrono>
namespace cg = cooperative_groups;
__global__ __launch_bounds__(1024) void coop_k(int* p, int nsync)
{
    cg::grid_group g = cg::this_grid();
    for (int i = 0; i < nsync; i++) { if (threadIdx.x == 0 && blockIdx.x == 0) p[i] = i; g.sync(); }
}
// hand-written barrier on a global counter (monotone target), self-resetting at the end
__global__ __launch_bounds__(1024) void spin_k(int* p, unsigned* bar, int nsync)
{
    const unsigned nb = gridDim.x;
    for (int i = 0; i < nsync; i++) {
        if (threadIdx.x == 0 && blockIdx.x == 0) p[i] = i;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            atomicAdd(bar, 1u);
            while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < nb * (i + 1)) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { if (atomicAdd(bar + 1, 1u) == nb - 1) { bar[0] = 0; bar[1] = 0; } }
}
__global__ void plain_k(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = 1; }
int main()
{
    int* p; unsigned* bar; hipMalloc(&p, 4096); hipMalloc(&bar, 64); hipMemset(bar, 0, 64);
    hipStream_t st; hipStreamCreate(&st);
    int nsync = 4; void* args[] = {&p, &nsync};
    for (int blocks : {64, 128, 256}) {
        for (int w = 0; w < 3; w++) hipLaunchCooperativeKernel((void*)coop_k, dim3(blocks), dim3(1024), args, 0, st);
        hipStreamSynchronize(st);
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < 200; i++) hipLaunchCooperativeKernel((void*)coop_k, dim3(blocks), dim3(1024), args, 0, st);
        hipStreamSynchronize(st);
        double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / 200;
        printf("coop  blocks=%d: %.2f us per launch (4 grid syncs)  err=%d\n", blocks, us, (int)hipGetLastError());
        for (int w = 0; w < 3; w++) hipLaunchKernelGGL(spin_k, dim3(blocks), dim3(1024), 0, st, p, bar, nsync);
        hipStreamSynchronize(st);
        t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < 200; i++) hipLaunchKernelGGL(spin_k, dim3(blocks), dim3(1024), 0, st, p, bar, nsync);
        hipStreamSynchronize(st);
        us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / 200;
        printf("spin  blocks=%d: %.2f us per launch (4 barriers)\n", blocks, us);
    }
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(plain_k, dim3(64), dim3(1024), 0, st, p);
    hipStreamSynchronize(st);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < 200; i++) for (int k = 0; k < 5; k++) hipLaunchKernelGGL(plain_k, dim3(64), dim3(1024), 0, st, p);
    hipStreamSynchronize(st);
    double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / 200;
    printf("plain: %.2f us per 5 launches\n", us);
    return 0;
}
