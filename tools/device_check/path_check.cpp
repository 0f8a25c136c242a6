// The bit-exact kernels of the path — K-nearest-neighbour search (grid kernels + tie replay), queryandgroup, furthest point sampling (resident and bucketed kernels),
// grid subsampling, sorted radius search — on fixed pseudo-random clouds with a lattice patch (exactly tied distances), outputs written as raw bytes.  Built twice
// from this one file (hipcc against libcbl_amd.so; g++ -DHOST_EMULATED against the host-emulated build of the same sources) and compared with `cmp`: the device
// and the emulator the CPU tests rest on must agree byte for byte.
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

static unsigned long long state = 0x9E3779B97F4A7C15ull;
static unsigned rnd() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return (unsigned)(state >> 11); }
static float unif() { return (rnd() & 0xffffff) / 16777216.0f; }
template <class T> static void dump(FILE* f, const char* what, const std::vector<T>& v)
{
    unsigned long long h = 1469598103934665603ull;
    const unsigned char* p = reinterpret_cast<const unsigned char*>(v.data());
    for (size_t i = 0; i < v.size() * sizeof(T); i++) { h ^= p[i]; h *= 1099511628211ull; }
    std::fwrite(v.data(), sizeof(T), v.size(), f);
    std::printf("%-28s %10zu bytes  fnv %016llx\n", what, v.size() * sizeof(T), h);
}

int main(int argc, char** argv)
{
    FILE* out = std::fopen(argc > 1 ? argv[1] : "path_check.bin", "wb");
    if (!out) return 2;
    // two clouds (2600 + 900 points); the first holds a 4 x 4 x 4 lattice patch: rows with exactly tied distances take the replay
    const int b = 2, n = 3500;
    std::vector<float> xyz((size_t)n * 3);
    for (auto& v : xyz) v = unif();
    for (int i = 0; i < 64; i++) { xyz[3 * (100 + i) + 0] = 0.3f + 0.0625f * (i & 3); xyz[3 * (100 + i) + 1] = 0.3f + 0.0625f * ((i >> 2) & 3); xyz[3 * (100 + i) + 2] = 0.3f + 0.0625f * (i >> 4); }
    for (int i = 2600; i < n; i++) xyz[3 * i] += 2.5f;
    std::vector<int> off = {2600, 3500};
    float* dxyz = dev(xyz); int* doff = dev(off);
    for (int K : {16, 36}) {
        std::vector<int> idx((size_t)n * K); std::vector<float> d2((size_t)n * K);
        int* didx = dev(idx); float* dd2 = dev(d2);
        const size_t wsb = cbl_knnquery_workspace_bytes(b, n, n, K);
        void* ws = scratch(wsb);
        RC(cbl_knnquery(b, n, n, K, dxyz, dxyz, doff, doff, didx, dd2, ws, wsb, nullptr));
        back(idx, didx); back(d2, dd2);
        dump(out, K == 16 ? "knn K=16 idx" : "knn K=36 idx", idx); dump(out, K == 16 ? "knn K=16 dist2" : "knn K=36 dist2", d2);
        if (K == 16) {                                                // the north-star gather on that table
            const int c = 32;
            std::vector<float> feat((size_t)n * c); for (auto& v : feat) v = unif() - 0.5f;
            std::vector<float> grouped((size_t)n * K * (3 + c));
            float *dfeat = dev(feat), *dg = dev(grouped);
            RC(cbl_queryandgroup(n, K, c, 1, dxyz, dxyz, dfeat, didx, dg, nullptr));
            back(grouped, dg); dump(out, "queryandgroup (n,16,35)", grouped);
        }
    }
    {   // sampler: both clouds to a quarter (the first through the resident kernel: below 3072 points), then ONE cloud of 3500 through the bucketed kernel
        std::vector<int> noff = {650, 875}, sidx(875); std::vector<float> tmp(n, 1e10f);
        int *dnoff = dev(noff), *dsidx = dev(sidx); float* dtmp = dev(tmp);
        const size_t wsb = cbl_furthestsampling_workspace_bytes(b, n, 2600); void* ws = scratch(wsb);
        RC(cbl_furthestsampling_ws(b, n, 2600, dxyz, doff, dnoff, dtmp, dsidx, ws, wsb, nullptr));
        back(sidx, dsidx); dump(out, "fps 2 clouds", sidx);
        std::vector<int> one = {n}, none = {875}; int *done = dev(one), *dnone = dev(none);
        std::vector<float> tmp2(n, 1e10f); float* dtmp2 = dev(tmp2);
        const size_t wsb2 = cbl_furthestsampling_workspace_bytes(1, n, n); void* ws2 = scratch(wsb2);
        RC(cbl_furthestsampling_ws(1, n, n, dxyz, done, dnone, dtmp2, dsidx, ws2, wsb2, nullptr));
        back(sidx, dsidx); dump(out, "fps 1 cloud (bucketed)", sidx);
    }
    {   // TF side: grid subsampling, then the sorted radius search of the sub-sampled points in the cloud
        std::vector<int> lens = {2600, 900};
        std::vector<float> sub((size_t)n * 3, 0.f); std::vector<int> slen(b), total(1);
        float* dsub = dev(sub); int *dslen = dev(slen), *dtotal = dev(total);
        const size_t wsb = cbl_grid_subsampling_workspace_bytes(b, n); void* ws = scratch(wsb);
        RC(cbl_grid_subsampling(b, n, dxyz, doff, 0.12f, 0, nullptr, 0, nullptr, dsub, nullptr, nullptr, dslen, dtotal, ws, wsb, nullptr));
        back(sub, dsub); back(slen, dslen); back(total, dtotal);
        const int m = total[0];
        sub.resize((size_t)m * 3); dump(out, "grid subsampling points", sub); dump(out, "grid subsampling lengths", slen);
        std::vector<int> soff = {slen[0], slen[0] + slen[1]}; int* dsoff = dev(soff);
        const int limit = 33;
        std::vector<int> nb((size_t)m * limit), counts(m), mx(1);
        int *dnb = dev(nb), *dcounts = dev(counts), *dmx = dev(mx);
        const size_t wsr = cbl_radius_neighbors_workspace_bytes(b, n); void* wr = scratch(wsr);
        RC(cbl_radius_neighbors(b, m, n, dsub, dxyz, dsoff, doff, 0.2f, limit, dnb, dcounts, dmx, wr, wsr, nullptr));
        back(nb, dnb); back(counts, dcounts); back(mx, dmx);
        dump(out, "radius neighbours", nb); dump(out, "radius counts", counts); dump(out, "radius max count", mx);
    }
    std::fclose(out);
    std::printf("PATH_CHECK_DONE\n");
    return 0;
}
