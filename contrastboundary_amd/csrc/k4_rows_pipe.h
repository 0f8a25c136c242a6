// K4 as a gather over the transposed neighbour table, for rows that are a slice of wider rows: the kernel in a header of its own so that the SAME source is
// also compiled for the host and run with wave semantics by tests/test_k4_rows_host.py (tests/host_emul/wave provides <hip/hip_runtime.h> and
// <gather_wave.h> for that build).  Included by neighbor_transpose.hip inside its anonymous namespace.
//   grouping backward   /root/reference/pytorch/lib/pointops/src/grouping/grouping_cuda_kernel.cu:16-25
#pragma once
#include <gather_wave.h>                                          // angle brackets: the host build of the tests puts its stand-in first on the include path

constexpr int K4_ROWS_BLOCK = 256;

// rows that are a SLICE of wider rows (the feature part of queryandgroup's (m, nsample, 3 + c) gradient: stride 3 + c floats, offset 3, so
// no 16-byte alignment): one wave per target, lane = channel, a row is one 4*c-byte contiguous request; eight rows in flight
__global__ __launch_bounds__(K4_ROWS_BLOCK) void grouping_bwd_csr_rows_kernel(unsigned n, int c, int stride, int off, const float* __restrict__ go,
                                                                   const int* __restrict__ order, const int* __restrict__ inv_start,
                                                                   const int* __restrict__ inv_src, float* __restrict__ gi)
{
    const unsigned lane = threadIdx.x & 63, wave = (unsigned)gw_uniform((int)(threadIdx.x >> 6));      // wave-uniform: scalar loads, uniform loops
    const unsigned nwg = (n + 3) >> 2;
    // A target is three dependent round trips (list bounds -> pair ids -> rows), and only during the third are rows in flight: with the
    // bounds fetched two targets ahead and the first 64 pair ids one target ahead (lane e holds entry e), a wave waits for rows only, and the
    // next batch of eight rows is requested before the previous one is summed.  The sum itself stays in ascending pair order (bit-exact).
    const unsigned vend = 8 * cbl_xcd_per(nwg), vstep = gridDim.x;
    auto bounds = [&](unsigned v, int& ok, unsigned& t, int& b0, int& b1) {
        const unsigned r = (v < vend ? cbl_xcd_slot(v, nwg) : 0u) * 4 + wave;
        ok = (v < vend && r < n) ? 1 : 0;
        const unsigned rc = ok ? r : 0u;
        b0 = inv_start[rc]; b1 = inv_start[rc + 1]; t = order ? (unsigned)order[rc] : rc;
    };
    auto ids = [&](int ok, int b0, int b1) -> int { const int e = b0 + (int)lane; return inv_src[(ok && e < b1) ? e : 0]; };
    int okA, s0A, s1A, okB, s0B, s1B, pB; unsigned tA, tB;
    bounds(blockIdx.x, okB, tB, s0B, s1B);
    pB = ids(okB, s0B, s1B);
    bounds(blockIdx.x + vstep, okA, tA, s0A, s1A);
    for (unsigned v = blockIdx.x; v < vend; v += vstep) {
        const int ok = okB, s0 = s0B, s1 = s1B, p0 = pB; const unsigned t = tB;
        okB = okA; tB = tA; s0B = s0A; s1B = s1A;
        pB = ids(okB, s0B, s1B);
        bounds(v + 2 * vstep, okA, tA, s0A, s1A);
        if (!ok) continue;
        for (unsigned ch = lane; ch < (unsigned)c; ch += 64) {
            const float* __restrict__ col = go + off + ch;
            float acc = 0.f;
            for (int eb = s0; eb < s1; eb += 64) {
                const int pl = eb == s0 ? p0 : inv_src[min(eb + (int)lane, s1 - 1)];
                const int cnt = min(64, s1 - eb);
                // entries past the end of the list re-read its last row and are not added
                auto load8 = [&](float (&x)[8], int u0) {
#pragma unroll
                    for (int u = 0; u < 8; u++) x[u] = col[(size_t)gw_readlane(pl, min(u0 + u, cnt - 1)) * stride];
                };
                auto add8 = [&](const float (&x)[8], int u0) {
#pragma unroll
                    for (int u = 0; u < 8; u++) acc = (u0 + u < cnt) ? acc + x[u] : acc;
                };
                float xa[8], xb[8];
                load8(xa, 0);
                for (int u0 = 0; u0 < cnt; u0 += 16) {
                    const bool second = u0 + 8 < cnt;
                    if (second) load8(xb, u0 + 8);
                    add8(xa, u0);
                    if (second) {
                        if (u0 + 16 < cnt) load8(xa, u0 + 16);
                        add8(xb, u0 + 8);
                    }
                }
            }
            gi[(size_t)t * c + ch] = acc;
        }
    }
}
