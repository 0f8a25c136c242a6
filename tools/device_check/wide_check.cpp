// The Point Transformer layer at the WIDE stages (cbl_pt_layer_wide_forward / _backward: the p chain and the narrow work by pt_layer.hip, the C-wide passes by the
// cbl_attn_* kernels of attention.hip) at (n, K, C) = (1280, 16, 128) and (640, 16, 256) on fixed pseudo-random inputs — once on the device, once under host
// emulation (same file, -DHOST_EMULATED), outputs compared by float_compare.py.  The feature gradients are scattered with float atomics here: their summation order
// differs from run to run on the device.
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
static unsigned long long state = 0xA0761D6478BD642Full;
static unsigned rnd() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return (unsigned)(state >> 11); }
static float unif() { return (rnd() & 0xffffff) / 16777216.0f; }
static std::vector<float> vec(size_t n, float scale, float shift = 0.f) { std::vector<float> v(n); for (auto& x : v) x = (unif() - 0.5f) * scale + shift; return v; }
static void dump(FILE* f, const char* what, const std::vector<float>& v)
{
    const unsigned n = (unsigned)v.size(); char name[32] = {0}; std::strncpy(name, what, 31);
    std::fwrite(name, 1, 32, f); std::fwrite(&n, 4, 1, f); std::fwrite(v.data(), 4, n, f);
    double s = 0; for (float x : v) s += x < 0 ? -x : x;
    std::printf("%-24s %8u floats  sum|x| %.9g\n", what, n, s);
}

int main(int argc, char** argv)
{
    FILE* out = std::fopen(argc > 1 ? argv[1] : "wide_check.bin", "wb");
    if (!out) return 2;
    const int nmax = 1280;
    const int n0 = nmax;
    std::vector<float> xyz((size_t)n0 * 3); for (auto& v : xyz) v = unif();
    float* dxyz = dev(xyz);
    for (int cfg = 0; cfg < 2; cfg++) {
        const int n = cfg ? 640 : 1280, K = 16, C = cfg ? 256 : 128, G = C / 8;
        std::vector<int> off = {n}; int* doff = dev(off);
        std::vector<int> idx((size_t)n * K), order(n); std::vector<float> d2((size_t)n * K);
        int *didx = dev(idx), *dorder = dev(order); float* dd2 = dev(d2);
        const size_t wsb = cbl_knnquery_workspace_bytes(1, n, n, K); void* ws = scratch(wsb);
        RC(cbl_knnquery(1, n, n, K, dxyz, dxyz, doff, doff, didx, dd2, ws, wsb, nullptr)); (void)dorder;
        // inputs and the fourteen parameters in the reference's layouts (include/cbl_amd.h: cbl_pt_layer_forward)
        float *xq = dev(vec((size_t)n * C, 2.f)), *xk = dev(vec((size_t)n * C, 2.f)), *xv = dev(vec((size_t)n * C, 2.f)), *gout = dev(vec((size_t)n * C, 2.f));
        float *Wp = dev(vec(9, 2.f)), *bp = dev(vec(3, 1.f)), *gp = dev(vec(3, 0.5f, 1.f)), *bep = dev(vec(3, 0.5f));
        float *W3C = dev(vec((size_t)C * 3, 1.f)), *b3C = dev(vec(C, 0.5f)), *gc = dev(vec(C, 0.5f, 1.f)), *bec = dev(vec(C, 0.5f));
        float *Wa = dev(vec((size_t)G * C, 0.5f)), *ba = dev(vec(G, 0.5f)), *gg = dev(vec(G, 0.5f, 1.f)), *beg = dev(vec(G, 0.5f)), *Wb = dev(vec((size_t)G * G, 1.f)), *bb = dev(vec(G, 0.5f));
        std::vector<float> p_r((size_t)n * K * 3), p0(p_r.size()), p1(p_r.size()), w2((size_t)n * K * G), a(w2.size()), o((size_t)n * C), consts(cbl_pt_layer_wide_consts_floats()), bnc((size_t)2 * C);
        float *dpr = dev(p_r), *dp0 = dev(p0), *dp1 = dev(p1), *dw2 = dev(w2), *da = dev(a), *dout = dev(o), *dconsts = dev(consts), *dbnc = dev(bnc);
        const size_t lb = cbl_pt_layer_wide_workspace_bytes(n, K, C); void* lws = scratch(lb);
        const float eps3[3] = {1e-5f, 1e-5f, 1e-5f}, mom3[3] = {0.1f, 0.1f, 0.1f};
        std::vector<float> rm3(3), rmc(C), rmg(G), rv3(3, 1.f), rvc(C, 1.f), rvg(G, 1.f); std::vector<long long> nb1(1);
        float* rmean[3] = {dev(rm3), dev(rmc), dev(rmg)}; float* rvar[3] = {dev(rv3), dev(rvc), dev(rvg)}; long long* nbt[3] = {dev(nb1), dev(nb1), dev(nb1)};
        RC(cbl_pt_layer_wide_forward(n, K, C, dxyz, xq, xk, xv, didx, Wp, bp, gp, bep, W3C, b3C, gc, bec, Wa, ba, gg, beg, Wb, bb, eps3, mom3, rmean, rvar, nbt,
                                dpr, dp0, dp1, dw2, da, dout, dconsts, dbnc, lws, lb, nullptr));
        back(o, dout); back(a, da); back(rvc, rvar[1]);
        dump(out, cfg ? "C256 out" : "C128 out", o); dump(out, cfg ? "C256 attention weights" : "C128 attention weights", a); dump(out, cfg ? "C256 running var (BN_c)" : "C128 running var (BN_c)", rvc);
        std::vector<float> gxq((size_t)n * C), gxk(gxq.size()), gxv(gxq.size()), gWp(9), gbp(3), ggp(3), gbep(3), gW3C((size_t)C * 3), gb3C(C), ggc(C), gbec(C), gWa((size_t)G * C), gba(G), ggg(G), gbeg(G), gWb((size_t)G * G), gbb(G);
        float *d1 = dev(gxq), *d2_ = dev(gxk), *d3 = dev(gxv), *e1 = dev(gWp), *e2 = dev(gbp), *e3 = dev(ggp), *e4 = dev(gbep), *e5 = dev(gW3C), *e6 = dev(gb3C), *e7 = dev(ggc), *e8 = dev(gbec),
              *e9 = dev(gWa), *e10 = dev(gba), *e11 = dev(ggg), *e12 = dev(gbeg), *e13 = dev(gWb), *e14 = dev(gbb);
        RC(cbl_pt_layer_wide_backward(n, K, C, xq, xk, xv, didx, gp, W3C, b3C, gc, bec, Wa, gg, Wb, dpr, dp0, dp1, dw2, da, dconsts, dbnc, gout, d1, d2_, d3, e1, e2, e3, e4, e5, e6, e7, e8,
                                 e9, e10, e11, e12, e13, e14, lws, lb, nullptr));
        back(gxq, d1); back(gxk, d2_); back(gxv, d3); back(gW3C, e5); back(gWa, e9); back(ggc, e7); back(gWp, e1);
        dump(out, cfg ? "C256 grad x_q" : "C128 grad x_q", gxq); dump(out, cfg ? "C256 grad x_k" : "C128 grad x_k", gxk); dump(out, cfg ? "C256 grad x_v" : "C128 grad x_v", gxv);
        dump(out, cfg ? "C256 grad W3C" : "C128 grad W3C", gW3C); dump(out, cfg ? "C256 grad Wa" : "C128 grad Wa", gWa); dump(out, cfg ? "C256 grad gamma_c" : "C128 grad gamma_c", ggc);
        dump(out, cfg ? "C256 grad Wp" : "C128 grad Wp", gWp);
    }
    std::fclose(out);
    std::printf("WIDE_CHECK_DONE\n");
    return 0;
}
