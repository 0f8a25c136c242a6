// Python-free timing + agreement of cbl_triple_linear_backward (the q / k / v projections' backward, blocks.py:33) for builds of the library given by path (dlopen):
// rows = 40960, C = 64 and 32, HIP events over `reps` calls; the weight gradients of the first library are the reference the others are compared with.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "hip error %d at line %d\n", (int)e_, __LINE__); std::exit(2); } } while (0)
typedef size_t (*ws_t)(int);
typedef int (*bwd_t)(long long, int, const float*, const float* const*, const float* const*, float*, float* const*, float* const*, void*, size_t, void*);
static unsigned long long state = 0x9FB21C651E98DF25ull;
static unsigned rnd() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return (unsigned)(state >> 11); }
static float* devrand(size_t n) { std::vector<float> h(n); for (auto& v : h) v = ((rnd() & 0xffff) / 65536.0f - 0.5f); float* d; CHECK(hipMalloc(&d, n * 4 + 64)); CHECK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice)); return d; }

int main(int argc, char** argv)
{
    const long long rows = 40960; const int reps = 200;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int C : {64, 32}) {
        float* x = devrand((size_t)rows * C);
        const float* w3[3]; const float* gy3[3]; float* gw3[3]; float* gb3[3];
        for (int p = 0; p < 3; p++) { w3[p] = devrand((size_t)C * C); gy3[p] = devrand((size_t)rows * C); CHECK(hipMalloc(&gw3[p], (size_t)C * C * 4)); CHECK(hipMalloc(&gb3[p], C * 4)); }
        float* gx; CHECK(hipMalloc(&gx, (size_t)rows * C * 4));
        std::vector<float> ref;
        for (int round = 0; round < 2; round++)
            for (int a = 1; a < argc; a++) {
                void* h = dlopen(argv[a], RTLD_NOW | RTLD_LOCAL);
                if (!h) { std::fprintf(stderr, "dlopen %s: %s\n", argv[a], dlerror()); return 2; }
                ws_t wsf = (ws_t)dlsym(h, "cbl_triple_linear_workspace_bytes"); bwd_t bwd = (bwd_t)dlsym(h, "cbl_triple_linear_backward");
                const size_t wb = wsf(C); void* ws; CHECK(hipMalloc(&ws, wb + 64));
                for (int w = 0; w < 20; w++) if (bwd(rows, C, x, w3, gy3, gx, gw3, gb3, ws, wb, nullptr)) return 3;
                CHECK(hipDeviceSynchronize()); CHECK(hipEventRecord(e0, nullptr));
                for (int r = 0; r < reps; r++) bwd(rows, C, x, w3, gy3, gx, gw3, gb3, ws, wb, nullptr);
                CHECK(hipEventRecord(e1, nullptr)); CHECK(hipEventSynchronize(e1));
                float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
                std::vector<float> g((size_t)C * C * 3 + 3 * C);
                for (int p = 0; p < 3; p++) { CHECK(hipMemcpy(g.data() + (size_t)p * C * C, gw3[p], (size_t)C * C * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(g.data() + (size_t)3 * C * C + p * C, gb3[p], C * 4, hipMemcpyDeviceToHost)); }
                double err = 0, scale = 0;
                if (ref.empty()) ref = g;
                for (size_t i = 0; i < g.size(); i++) { err = std::fmax(err, std::fabs((double)g[i] - ref[i])); scale = std::fmax(scale, std::fabs((double)ref[i])); }
                std::printf("C %2d %-46s round %d: %.2f us per backward (input + weight gradients), weight / bias gradients vs the first library: %.2e of their scale\n", C, argv[a], round, ms * 1000.f / reps, err / scale);
                CHECK(hipFree(ws));
            }
    }
    return 0;
}
