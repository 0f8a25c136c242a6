// TEST INFRASTRUCTURE (not product code): the product's bucketed furthest point sampling (contrastboundary_amd/csrc/fps_bucket.hip + fps_wave.h) compiled for the
// HOST with wave semantics (tests/host_emul/wave: every thread a fibre; DPP / readlane / ballot / barriers as rendezvous; rocprim's sort as std::stable_sort).
// tests/test_fps_bucket_host.py holds its sample sequences against the oracle bit for bit, without a GPU.
#include "amdgcn.h"

// knn_grid.hip's bounding-box pass, restated for the host: per cloud the order-preserving keys of min (slots 0..2) and max (3..5); the caller pre-set 0xffffffff / 0
static unsigned host_f2key(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
int cbl_bbox_keys_launch(int b, int n, const float* xyz, const int* offset, unsigned* bbox, hipStream_t)
{
    for (int c = 0; c < b; c++) {
        const int s = c ? offset[c - 1] : 0, e = offset[c] < n ? offset[c] : n;
        for (int i = s; i < e; i++)
            for (int a = 0; a < 3; a++) {
                const unsigned k = host_f2key(xyz[3 * i + a]);
                if (k < bbox[6 * c + a]) bbox[6 * c + a] = k;
                if (k > bbox[6 * c + 3 + a]) bbox[6 * c + 3 + a] = k;
            }
    }
    return 0;
}

#include "../../contrastboundary_amd/csrc/fps_bucket.hip"

extern "C" size_t host_fps_bucket_workspace_bytes(int b, int n) { return cbl_fps_bucket_workspace_bytes(b, n); }
extern "C" int host_fps_bucket(int b, int n, int n_max, int bits, const float* xyz, const int* offset, const int* new_offset, float* tmp, int* idx,
                               void* ws, size_t ws_bytes, const int* prefix_cert, int* cert_out)
{
    emul::S().stack_bytes = 256 * 1024;                             // the sample loop's frames (lambdas over register arrays) are larger than the attention layer's
    return cbl_fps_bucket_launch(b, n, n_max, bits, xyz, offset, new_offset, tmp, idx, ws, ws_bytes, nullptr, prefix_cert, cert_out);
}
