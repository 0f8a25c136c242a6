// The fused Point Transformer layer (pt_layer.hip: /root/reference/pytorch/model/blocks.py:31-44 as one pass structure, train-mode BatchNorms) forward and backward
// at (n, K, C) = (3000, 16, 64) and (3000, 8, 32) on fixed pseudo-random inputs, with the search's cell order as processing order — once on the device, once
// under host emulation (same file, -DHOST_EMULATED), outputs compared by float_compare.py.  The layer's sums run in a fixed order on both.
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
    FILE* out = std::fopen(argc > 1 ? argv[1] : "layer_check.bin", "wb");
    if (!out) return 2;
    const int n = 3000;
    std::vector<float> xyz((size_t)n * 3); for (auto& v : xyz) v = unif();
    std::vector<int> off = {n};
    float* dxyz = dev(xyz); int* doff = dev(off);
    for (int cfg = 0; cfg < 2; cfg++) {
        const int K = cfg ? 8 : 16, C = cfg ? 32 : 64, G = C / 8;
        std::vector<int> idx((size_t)n * K), order(n); std::vector<float> d2((size_t)n * K);
        int *didx = dev(idx), *dorder = dev(order); float* dd2 = dev(d2);
        const size_t wsb = cbl_knnquery_workspace_bytes(1, n, n, K); void* ws = scratch(wsb);
        RC(cbl_knnquery_ordered(1, n, n, K, dxyz, dxyz, doff, doff, didx, dd2, 0, dorder, ws, wsb, nullptr));
        std::vector<int> inv_start(n + 1), inv_src((size_t)n * K);
        int *dis = dev(inv_start), *dsrc = dev(inv_src);
        const size_t tb = cbl_neighbor_transpose_workspace_bytes(n, n, K); void* tws = scratch(tb);
        RC(cbl_neighbor_transpose(n, n, K, didx, dorder, dorder, dis, dsrc, tws, tb, nullptr));
        // inputs and the fourteen parameters in the reference's layouts (include/cbl_amd.h: cbl_pt_layer_forward)
        float *xq = dev(vec((size_t)n * C, 2.f)), *xk = dev(vec((size_t)n * C, 2.f)), *xv = dev(vec((size_t)n * C, 2.f)), *gout = dev(vec((size_t)n * C, 2.f));
        float *Wp = dev(vec(9, 2.f)), *bp = dev(vec(3, 1.f)), *gp = dev(vec(3, 0.5f, 1.f)), *bep = dev(vec(3, 0.5f));
        float *W3C = dev(vec((size_t)C * 3, 1.f)), *b3C = dev(vec(C, 0.5f)), *gc = dev(vec(C, 0.5f, 1.f)), *bec = dev(vec(C, 0.5f));
        float *Wa = dev(vec((size_t)G * C, 0.5f)), *ba = dev(vec(G, 0.5f)), *gg = dev(vec(G, 0.5f, 1.f)), *beg = dev(vec(G, 0.5f)), *Wb = dev(vec((size_t)G * G, 1.f)), *bb = dev(vec(G, 0.5f));
        std::vector<float> p_r((size_t)n * K * 3), p0(p_r.size()), p1(p_r.size()), w2((size_t)n * K * G), a(w2.size()), o((size_t)n * C), consts(cbl_pt_layer_consts_floats());
        float *dpr = dev(p_r), *dp0 = dev(p0), *dp1 = dev(p1), *dw2 = dev(w2), *da = dev(a), *dout = dev(o), *dconsts = dev(consts);
        const size_t lb = cbl_pt_layer_workspace_bytes(n, K, C); void* lws = scratch(lb);
        const float eps3[3] = {1e-5f, 1e-5f, 1e-5f}, mom3[3] = {0.1f, 0.1f, 0.1f};
        std::vector<float> rm3(3), rmc(C), rmg(G), rv3(3, 1.f), rvc(C, 1.f), rvg(G, 1.f); std::vector<long long> nb1(1);
        float* rmean[3] = {dev(rm3), dev(rmc), dev(rmg)}; float* rvar[3] = {dev(rv3), dev(rvc), dev(rvg)}; long long* nbt[3] = {dev(nb1), dev(nb1), dev(nb1)};
        RC(cbl_pt_layer_forward(n, K, C, dxyz, xq, xk, xv, didx, dorder, Wp, bp, gp, bep, W3C, b3C, gc, bec, Wa, ba, gg, beg, Wb, bb, eps3, mom3, rmean, rvar, nbt,
                                dpr, dp0, dp1, dw2, da, dout, dconsts, lws, lb, nullptr));
        back(o, dout); back(a, da); back(rvc, rvar[1]);
        dump(out, cfg ? "C32 out" : "C64 out", o); dump(out, cfg ? "C32 attention weights" : "C64 attention weights", a); dump(out, cfg ? "C32 running var (BN_c)" : "C64 running var (BN_c)", rvc);
        std::vector<float> gxq((size_t)n * C), gxk(gxq.size()), gxv(gxq.size()), gWp(9), gbp(3), ggp(3), gbep(3), gW3C((size_t)C * 3), gb3C(C), ggc(C), gbec(C), gWa((size_t)G * C), gba(G), ggg(G), gbeg(G), gWb((size_t)G * G), gbb(G);
        float *d1 = dev(gxq), *d2_ = dev(gxk), *d3 = dev(gxv), *e1 = dev(gWp), *e2 = dev(gbp), *e3 = dev(ggp), *e4 = dev(gbep), *e5 = dev(gW3C), *e6 = dev(gb3C), *e7 = dev(ggc), *e8 = dev(gbec),
              *e9 = dev(gWa), *e10 = dev(gba), *e11 = dev(ggg), *e12 = dev(gbeg), *e13 = dev(gWb), *e14 = dev(gbb);
        RC(cbl_pt_layer_backward(n, K, C, xq, xk, xv, didx, dorder, dis, dsrc, gp, W3C, b3C, gc, Wa, gg, Wb, dpr, dp0, dp1, dw2, da, dconsts, gout, d1, d2_, d3, e1, e2, e3, e4, e5, e6, e7, e8,
                                 e9, e10, e11, e12, e13, e14, lws, lb, nullptr));
        back(gxq, d1); back(gxk, d2_); back(gxv, d3); back(gW3C, e5); back(gWa, e9); back(ggc, e7); back(gWp, e1);
        dump(out, cfg ? "C32 grad x_q" : "C64 grad x_q", gxq); dump(out, cfg ? "C32 grad x_k" : "C64 grad x_k", gxk); dump(out, cfg ? "C32 grad x_v" : "C64 grad x_v", gxv);
        dump(out, cfg ? "C32 grad W3C" : "C64 grad W3C", gW3C); dump(out, cfg ? "C32 grad Wa" : "C64 grad Wa", gWa); dump(out, cfg ? "C32 grad gamma_c" : "C64 grad gamma_c", ggc);
        dump(out, cfg ? "C32 grad Wp" : "C64 grad Wp", gWp);
    }
    std::fclose(out);
    std::printf("LAYER_CHECK_DONE\n");
    return 0;
}
