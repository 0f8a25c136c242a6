// Stand-alone check of the contrast kernels through the C ABI, without Python: fixed pseudo-random inputs, every flavour whose coefficient arithmetic changed
// (softnn, 'nce', margin 'S'; pytorch and TF heads; the scatter kernels of cbl.hip), results written as raw floats.  Built twice from this one file:
//   hipcc  -> runs on the GPU against contrastboundary_amd/lib/libcbl_amd.so          (tools/device_check/run.sh)
//   g++ -DHOST_EMULATED -> runs here against the host build of the same kernels        (tests/host_emul/full_library.py)
// tools/device_check/compare.py holds the two outputs against each other.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/cbl_amd.h"
#ifndef HOST_EMULATED
#include <hip/hip_runtime.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "hip error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); std::exit(2); } } while (0)
template <class T> T* dev(const std::vector<T>& h) { T* d; CHECK(hipMalloc(&d, h.size() * sizeof(T) + 16)); CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }
template <class T> void back(std::vector<T>& h, const T* d) { CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(h.data(), d, h.size() * sizeof(T), hipMemcpyDeviceToHost)); }
#else
template <class T> T* dev(const std::vector<T>& h) { T* d = (T*)std::aligned_alloc(64, (h.size() * sizeof(T) + 127) / 64 * 64); std::memcpy(d, h.data(), h.size() * sizeof(T)); return d; }
template <class T> void back(std::vector<T>& h, const T* d) { std::memcpy(h.data(), d, h.size() * sizeof(T)); }
#endif

static unsigned long long state = 88172645463325252ull;
static unsigned rnd() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return (unsigned)(state >> 11); }
static float unif() { return (rnd() & 0xffffff) / 16777216.0f; }

int main(int argc, char** argv)
{
    const char* out_path = argc > 1 ? argv[1] : "cbl_check.bin";
    FILE* out = std::fopen(out_path, "wb");
    if (!out) { std::fprintf(stderr, "cannot write %s\n", out_path); return 2; }
    const int n = 3000;
    int cases = 0;
    for (int nsample : {9, 17, 36}) for (int d : {16, 32}) for (int flags : {0, 1, 4, 5, 9, 13}) for (float scale : {0.4f, 4.0f}) {
        const bool tf = flags & 1;
        std::vector<float> feat((size_t)n * d);
        for (auto& v : feat) v = (unif() + unif() + unif() - 1.5f) * scale;
        std::vector<int> lab(n), idx((size_t)n * nsample);
        for (int i = 0; i < n; i++) lab[i] = (i / 40) % 5 == 4 && tf ? -1 : (int)(rnd() % 4);
        for (int i = 0; i < n; i++) for (int j = 0; j < nsample; j++) {
            int t = i + (int)(rnd() % 61) - 30;                       // neighbours nearby in index, some outside (shadow in the TF head)
            if (t < 0) t = tf ? n : 0;
            if (t >= n) t = tf ? n : n - 1;
            idx[(size_t)i * nsample + j] = j == 0 ? i : t;
        }
        std::vector<float> per_point(n), stats(2), loss(1), coef((size_t)n * nsample), own((size_t)n * d), grad((size_t)n * d, 0.f), one(1, 1.0f);
        std::vector<int> mask(n);
        float *dfeat = dev(feat), *dpp = dev(per_point), *dstats = dev(stats), *dloss = dev(loss), *dcoef = dev(coef), *down = dev(own), *dgrad = dev(grad), *done = dev(one);
        int *dlab = dev(lab), *didx = dev(idx), *dmask = dev(mask);
        const float T = 0.7f, weight = 0.1f;
        int rc = cbl_contrast_pairs_forward(n, tf ? n : 0x7fffffff, flags, nsample, d, dfeat, dlab, 0, 0.f, didx, nullptr, T, weight, dpp, dmask, dstats, dloss, dcoef, down, nullptr);
        if (rc) { std::fprintf(stderr, "forward rc %d (nsample %d d %d flags %d)\n", rc, nsample, d, flags); return 3; }
        rc = cbl_contrast_pairs_backward_atomic(n, n, nsample, d, dfeat, dcoef, down, didx, dstats, done, weight, dgrad, nullptr);
        if (rc) { std::fprintf(stderr, "backward rc %d\n", rc); return 3; }
        back(loss, dloss); back(stats, dstats); back(grad, dgrad); back(per_point, dpp); back(mask, dmask);
        const int head[4] = {nsample, d, flags, (int)(scale * 10)};
        std::fwrite(head, sizeof(int), 4, out); std::fwrite(loss.data(), 4, 1, out); std::fwrite(stats.data(), 4, 2, out);
        std::fwrite(per_point.data(), 4, n, out); std::fwrite(mask.data(), 4, n, out); std::fwrite(grad.data(), 4, grad.size(), out);
        double gs = 0; for (float v : grad) gs += v < 0 ? -v : v;
        std::printf("pairs nsample %2d d %2d flags %2d scale %.1f: loss %.7g count %g |grad| %.7g\n", nsample, d, flags, scale, loss[0], stats[1], gs);
        // the scatter kernels of cbl.hip (softnn only), both heads
        if (flags <= 1) {
            std::vector<float> g2((size_t)n * d, 0.f), st2(2), l2(1), pp2(n); std::vector<int> m2(n);
            float *dg2 = dev(g2), *dst2 = dev(st2), *dl2 = dev(l2), *dpp2 = dev(pp2); int* dm2 = dev(m2);
            if (tf) {
                rc = cbl_tf_contrast_forward(n, n, nsample, d, dfeat, dlab, didx, T, weight, dpp2, dm2, dst2, dl2, nullptr);
                if (!rc) rc = cbl_tf_contrast_backward(n, n, nsample, d, dfeat, dlab, didx, T, weight, dst2, done, dg2, nullptr);
            } else {
                rc = cbl_point_contrast_forward(n, nsample, d, dfeat, dlab, didx, T, weight, dpp2, dm2, dst2, dl2, nullptr);
                if (!rc) rc = cbl_point_contrast_backward(n, nsample, d, dfeat, dlab, didx, T, weight, dst2, done, dg2, nullptr);
            }
            if (rc) { std::fprintf(stderr, "scatter rc %d\n", rc); return 3; }
            back(l2, dl2); back(g2, dg2);
            std::fwrite(l2.data(), 4, 1, out); std::fwrite(g2.data(), 4, g2.size(), out);
            double gs2 = 0; for (float v : g2) gs2 += v < 0 ? -v : v;
            std::printf("  scatter kernels: loss %.7g |grad| %.7g\n", l2[0], gs2);
        }
        cases++;
    }
    std::fclose(out);
    std::printf("CBL_CHECK_DONE %d cases -> %s\n", cases, out_path);
    return 0;
}
