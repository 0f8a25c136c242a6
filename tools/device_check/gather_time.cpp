// Python-free timing of the north-star gather at the bench shape (N = 40960, K = 16, C = 64: 189.7 MB algorithmic per launch): the K = 16 self-search
// with its cell order, then `reps` launches of cbl_queryandgroup_ordered between HIP events — a second harness beside bench.py's `roofline` (which times the same
// entry inside the step).  Points: uniform in a 4 x 4 x 1 slab (a room's aspect), so the neighbour rows are as local as the bench scene's.
//   hipcc --offload-arch=gfx950 -O2 tools/device_check/gather_time.cpp -o tools/device_check/gather_time_dev -Lcontrastboundary_amd/lib -lcbl_amd
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <hip/hip_runtime.h>
#include "../../include/cbl_amd.h"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "hip error %d at line %d\n", (int)e_, __LINE__); std::exit(2); } } while (0)
#define RC(x) do { int rc_ = (x); if (rc_) { std::fprintf(stderr, "%s -> %d\n", #x, rc_); return 3; } } while (0)
static unsigned long long state = 0x2545F4914F6CDD1Dull;
static unsigned rnd() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return (unsigned)(state >> 11); }
static float unif() { return (rnd() & 0xffffff) / 16777216.0f; }
template <class T> T* dev(const std::vector<T>& h) { T* d; CHECK(hipMalloc(&d, h.size() * sizeof(T) + 64)); CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }

int main()
{
    const int n = 40960, K = 16, C = 64, reps = 300;
    std::vector<float> xyz((size_t)n * 3), feat((size_t)n * C);
    for (int i = 0; i < n; i++) { xyz[3 * i] = 4.f * unif(); xyz[3 * i + 1] = 4.f * unif(); xyz[3 * i + 2] = unif(); }
    for (auto& v : feat) v = unif() - 0.5f;
    std::vector<int> off = {n};
    float *dxyz = dev(xyz), *dfeat = dev(feat); int* doff = dev(off);
    int *didx, *dorder; float *dd2, *dout;
    CHECK(hipMalloc(&didx, (size_t)n * K * 4)); CHECK(hipMalloc(&dd2, (size_t)n * K * 4)); CHECK(hipMalloc(&dorder, n * 4)); CHECK(hipMalloc(&dout, (size_t)n * K * (3 + C) * 4));
    const size_t wsb = cbl_knnquery_workspace_bytes(1, n, n, K);
    void* ws; CHECK(hipMalloc(&ws, wsb + 64));
    RC(cbl_knnquery_ordered(1, n, n, K, dxyz, dxyz, doff, doff, didx, dd2, 0, dorder, ws, wsb, nullptr));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const double bytes = 12.0 * n + 12.0 * n + 4.0 * n * C + 4.0 * n * K + 4.0 * n * K * (3 + C);   // xyz, new_xyz, feat, idx read; out written
    for (int ordered = 1; ordered >= 0; ordered--)
        for (int round = 0; round < 3; round++) {
            for (int w = 0; w < 20; w++) RC(cbl_queryandgroup_ordered(n, K, C, 1, dxyz, dxyz, dfeat, didx, ordered ? dorder : nullptr, dout, nullptr));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, nullptr));
            for (int r = 0; r < reps; r++) cbl_queryandgroup_ordered(n, K, C, 1, dxyz, dxyz, dfeat, didx, ordered ? dorder : nullptr, dout, nullptr);
            CHECK(hipEventRecord(e1, nullptr)); CHECK(hipEventSynchronize(e1));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1000.0 / reps;
            std::printf("queryandgroup %s round %d: %.2f us per launch, %.1f MB algorithmic -> %.2f TB/s = %.3f of 8 TB/s\n", ordered ? "in cell order " : "in index order", round, us, bytes / 1e6,
                        bytes / us / 1e6, bytes / us / 1e6 / 8.0);
        }
    // the fill rate of this box for the same number of bytes written (hipMemsetAsync), the ceiling a write-dominated kernel can reach here
    CHECK(hipDeviceSynchronize()); CHECK(hipEventRecord(e0, nullptr));
    for (int r = 0; r < 100; r++) CHECK(hipMemsetAsync(dout, 0, (size_t)n * K * (3 + C) * 4, nullptr));
    CHECK(hipEventRecord(e1, nullptr)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("fill of the output (%.1f MB): %.2f us -> %.2f TB/s\n", (double)n * K * (3 + C) * 4 / 1e6, ms * 10.0, (double)n * K * (3 + C) * 4 / (ms * 10.0) / 1e6);
    return 0;
}
