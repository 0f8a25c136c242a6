// The matrix-core and float kernels that were NOT touched after the measured commit — KPConv forward (v_mfma_f32_16x16x4_f32) and its gather-form backward over the
// transposed table, AdaptiveWeight forward, a skinny Linear (MFMA), train-mode BatchNorm + ReLU — on fixed pseudo-random inputs, outputs written raw.  Built twice
// (hipcc against libcbl_amd.so / g++ -DHOST_EMULATED against the host-emulated build, where an MFMA is its k-ordered fmaf chain) and compared by float_compare.py:
// how close is the emulator the CPU tests rest on to the device where the arithmetic is not integer?
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/cbl_amd.h"
#ifndef HOST_EMULATED
#include <hip/hip_runtime.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "hip error %d at line %d\n", (int)e_, __LINE__); std::exit(2); } } while (0)
template <class T> T* dev(const std::vector<T>& h) { T* d; CHECK(hipMalloc(&d, h.size() * sizeof(T) + 64)); CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }
template <class T> void back(std::vector<T>& h, const T* d) { CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(h.data(), d, h.size() * sizeof(T), hipMemcpyDeviceToHost)); }
static void* scratch(size_t bytes) { void* d; CHECK(hipMalloc(&d, bytes + 64)); CHECK(hipMemset(d, 0, bytes + 64)); return d; }
#else
template <class T> T* dev(const std::vector<T>& h) { T* d = (T*)std::aligned_alloc(64, (h.size() * sizeof(T) + 127) / 64 * 64); std::memcpy(d, h.data(), h.size() * sizeof(T)); return d; }
template <class T> void back(std::vector<T>& h, const T* d) { std::memcpy(h.data(), d, h.size() * sizeof(T)); }
static void* scratch(size_t bytes) { void* d = std::aligned_alloc(64, (bytes + 127) / 64 * 64); std::memset(d, 0, (bytes + 127) / 64 * 64); return d; }
#endif
#define RC(x) do { int rc_ = (x); if (rc_) { std::fprintf(stderr, "%s -> %d\n", #x, rc_); return 3; } } while (0)
static unsigned long long state = 0xD1B54A32D192ED03ull;
static unsigned rnd() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return (unsigned)(state >> 11); }
static float unif() { return (rnd() & 0xffffff) / 16777216.0f; }
static void dump(FILE* f, const char* what, const std::vector<float>& v)
{
    const unsigned n = (unsigned)v.size(); char name[32] = {0}; std::strncpy(name, what, 31);
    std::fwrite(name, 1, 32, f); std::fwrite(&n, 4, 1, f); std::fwrite(v.data(), 4, n, f);
    double s = 0; for (float x : v) s += x < 0 ? -x : x;
    std::printf("%-24s %8u floats  sum|x| %.9g\n", what, n, s);
}

int main(int argc, char** argv)
{
    FILE* out = std::fopen(argc > 1 ? argv[1] : "float_check.bin", "wb");
    if (!out) return 2;
    const int n = 3000, K = 16, C = 64, KP = 15;
    std::vector<float> xyz((size_t)n * 3), feat((size_t)n * C), kp((size_t)KP * 3), kw((size_t)KP * C), go((size_t)n * C);
    for (auto& v : xyz) v = unif();
    for (auto& v : feat) v = unif() - 0.5f;
    for (auto& v : kp) v = (unif() - 0.5f) * 0.12f;
    kp[0] = kp[1] = kp[2] = 0.f;
    for (auto& v : kw) v = unif() - 0.5f;
    for (auto& v : go) v = unif() - 0.5f;
    std::vector<int> off = {n}, idx((size_t)n * K); std::vector<float> d2((size_t)n * K);
    float *dxyz = dev(xyz), *dfeat = dev(feat), *dkp = dev(kp), *dkw = dev(kw), *dgo = dev(go), *dd2 = dev(d2); int *doff = dev(off), *didx = dev(idx);
    const size_t wsb = cbl_knnquery_workspace_bytes(1, n, n, K); void* ws = scratch(wsb);
    RC(cbl_knnquery(1, n, n, K, dxyz, dxyz, doff, doff, didx, dd2, ws, wsb, nullptr));
    {   // KPConv forward ('linear' influence, 'sum'), then the gather-form backward over the transposed table
        std::vector<float> o((size_t)n * C), gf((size_t)n * C), gkw((size_t)KP * C);
        float *dout = dev(o), *dgf = dev(gf), *dgkw = dev(gkw);
        RC(cbl_kpconv_forward(n, n, K, C, KP, dxyz, dxyz, didx, dfeat, dkp, dkw, 0.09f, 1, 0, dout, nullptr));
        back(o, dout); dump(out, "kpconv forward", o);
        std::vector<int> inv_start(n + 1), inv_src((size_t)n * K);
        int *dis = dev(inv_start), *dsrc = dev(inv_src);
        const size_t tb = cbl_neighbor_transpose_workspace_bytes(n, n, K); void* tws = scratch(tb);
        RC(cbl_neighbor_transpose(n, n, K, didx, nullptr, nullptr, dis, dsrc, tws, tb, nullptr));
        const size_t kb = cbl_kpconv_backward_csr_workspace_bytes(n, C, KP); void* kws = scratch(kb);
        RC(cbl_kpconv_backward_csr(n, n, K, C, KP, dxyz, dxyz, dfeat, dkp, dkw, 0.09f, 1, 0, dgo, nullptr, dis, dsrc, dgf, dgkw, kws, kb, nullptr));
        back(gf, dgf); back(gkw, dgkw); dump(out, "kpconv grad features", gf); dump(out, "kpconv grad weights", gkw);
    }
    {   // AdaptiveWeight forward ('mean')
        std::vector<float> W((size_t)3 * C), b(C), o((size_t)n * C); for (auto& v : W) v = unif() - 0.5f; for (auto& v : b) v = unif() - 0.5f;
        std::vector<int> pad(1); float *dW = dev(W), *db = dev(b), *dout = dev(o); int* dpad = dev(pad);
        RC(cbl_index_max((long long)n * K, didx, dpad, nullptr));
        RC(cbl_adaptive_weight_forward(n, n, K, C, dxyz, dxyz, didx, dfeat, 0.1f, dW, db, dpad, 1, dout, nullptr));
        back(o, dout); dump(out, "adaptive weight forward", o);
    }
    {   // Linear 64 -> 32 on the matrix cores, then train-mode BatchNorm + ReLU over its rows
        const int co = 32;
        std::vector<float> W((size_t)co * C), b(co), y((size_t)n * co), g(co), be(co), z((size_t)n * co), sm(co), si(co);
        for (auto& v : W) v = (unif() - 0.5f) * 0.25f; for (auto& v : b) v = unif() - 0.5f; for (auto& v : g) v = 0.5f + unif(); for (auto& v : be) v = unif() - 0.5f;
        float *dW = dev(W), *db = dev(b), *dy = dev(y), *dg = dev(g), *dbe = dev(be), *dz = dev(z), *dsm = dev(sm), *dsi = dev(si);
        RC(cbl_skinny_linear_forward(n, C, co, dfeat, dW, db, dy, nullptr));
        back(y, dy); dump(out, "linear 64->32", y);
        const size_t bb = cbl_bn_rows_workspace_bytes(n, co); void* bws = scratch(bb);
        RC(cbl_bn_rows_forward(n, co, dy, dg, dbe, 1e-5f, 0.1f, nullptr, nullptr, nullptr, 1, dsm, dsi, dz, bws, bb, nullptr));
        back(z, dz); back(sm, dsm); back(si, dsi); dump(out, "batchnorm + relu", z); dump(out, "batch mean", sm); dump(out, "batch invstd", si);
    }
    std::fclose(out);
    std::printf("FLOAT_CHECK_DONE\n");
    return 0;
}
