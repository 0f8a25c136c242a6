// Python-free timing of the contrast head's pair kernel at the bench shape (n = 40960, K = 36, d = 32): HIP events around `reps` launches, for any build of the
// library given by path (dlopen) — the shipped one next to a previous state, in ONE process on ONE box.
//   hipcc --offload-arch=gfx950 -O2 tools/device_check/cbl_time.cpp -o tools/device_check/cbl_time_dev -ldl
//   ./tools/device_check/cbl_time_dev <libA.so> <libB.so> ...
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "hip error %d at line %d\n", (int)e_, __LINE__); std::exit(2); } } while (0)
typedef int (*fwd_t)(int, int, int, int, int, const float*, const void*, int, float, const int*, const int*, float, float, float*, int*, float*, float*, float*, float*, void*);
static unsigned long long state = 88172645463325252ull;
static unsigned rnd() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return (unsigned)(state >> 11); }
template <class T> T* dev(const std::vector<T>& h) { T* d; CHECK(hipMalloc(&d, h.size() * sizeof(T) + 16)); CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }

int main(int argc, char** argv)
{
    const int n = 40960, nsample = 36, d = 32, reps = 200;
    std::vector<float> feat((size_t)n * d); for (auto& v : feat) v = ((rnd() & 0xffff) / 65536.0f - 0.5f);
    std::vector<int> lab(n), idx((size_t)n * nsample);
    for (int i = 0; i < n; i++) lab[i] = (i / 97) % 13;               // blocky labels: a few per cent of the points are boundary points, as in a room
    for (int i = 0; i < n; i++) for (int j = 0; j < nsample; j++) { int t = i + (int)(rnd() % 401) - 200; t = t < 0 ? 0 : (t >= n ? n - 1 : t); idx[(size_t)i * nsample + j] = j ? t : i; }
    float *dfeat = dev(feat); int *dlab = dev(lab), *didx = dev(idx);
    float *dpp, *dstats, *dloss, *dcoef, *down; int* dmask;
    CHECK(hipMalloc(&dpp, n * 4)); CHECK(hipMalloc(&dstats, 8)); CHECK(hipMalloc(&dloss, 4)); CHECK(hipMalloc(&dcoef, (size_t)n * nsample * 4)); CHECK(hipMalloc(&down, (size_t)n * d * 4)); CHECK(hipMalloc(&dmask, n * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int round = 0; round < 2; round++)
        for (int a = 1; a < argc; a++) {
            void* h = dlopen(argv[a], RTLD_NOW | RTLD_LOCAL);
            if (!h) { std::fprintf(stderr, "dlopen %s: %s\n", argv[a], dlerror()); return 2; }
            fwd_t fwd = (fwd_t)dlsym(h, "cbl_contrast_pairs_forward");
            for (int grad = 0; grad < 2; grad++) {
                for (int w = 0; w < 20; w++) if (fwd(n, 0x7fffffff, 0, nsample, d, dfeat, dlab, 0, 0.f, didx, nullptr, 1.0f, 0.1f, dpp, dmask, dstats, dloss, grad ? dcoef : nullptr, grad ? down : nullptr, nullptr)) return 3;
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0, nullptr));
                for (int r = 0; r < reps; r++) fwd(n, 0x7fffffff, 0, nsample, d, dfeat, dlab, 0, 0.f, didx, nullptr, 1.0f, 0.1f, dpp, dmask, dstats, dloss, grad ? dcoef : nullptr, grad ? down : nullptr, nullptr);
                CHECK(hipEventRecord(e1, nullptr)); CHECK(hipEventSynchronize(e1));
                float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
                float loss = 0, stats[2]; CHECK(hipMemcpy(&loss, dloss, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(stats, dstats, 8, hipMemcpyDeviceToHost));
                std::printf("%-48s round %d %s: %.2f us per call (mining + finalize), loss %.7g count %g\n", argv[a], round, grad ? "with coefficients" : "loss only        ", ms * 1000.f / reps, loss, stats[1]);
            }
        }
    return 0;
}
