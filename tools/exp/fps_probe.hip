// standalone probe: which part of an FPS iteration costs what (compile: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CBL_EXPORT extern "C"
#include "../../contrastboundary_amd/csrc/fps.hip"

int main()
{
    for (int n : {10240, 40960}) {
        int m = n / 4;
        std::vector<float> h(3 * n); for (auto& v : h) v = rand() / (float)RAND_MAX;
        float *xyz, *tmp; int *off, *noff, *idx;
        hipMalloc(&xyz, 12 * n); hipMalloc(&tmp, 4 * n); hipMalloc(&off, 4); hipMalloc(&noff, 4); hipMalloc(&idx, 4 * m);
        hipMemcpy(xyz, h.data(), 12 * n, hipMemcpyHostToDevice); hipMemcpy(off, &n, 4, hipMemcpyHostToDevice); hipMemcpy(noff, &m, 4, hipMemcpyHostToDevice);
        std::vector<float> t(n, 1e10f);
        for (int rep = 0; rep < 2; rep++) {
            hipMemcpy(tmp, t.data(), 4 * n, hipMemcpyHostToDevice);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            cbl_furthestsampling(1, n, xyz, off, noff, tmp, idx, nullptr);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("n=%d m=%d: %.3f ms  (%.3f us / sample)\n", n, m, ms, ms * 1e3 / m);
        }
    }
    return 0;
}
