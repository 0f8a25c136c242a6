// TEST INFRASTRUCTURE: host stand-in for the one rocprim entry our kernels' host code calls (a stable sort of (key, value) pairs on a bit range of the key)
#pragma once
#include <hip/hip_runtime.h>
#include <numeric>
namespace rocprim {
template <class K, class V>
inline hipError_t radix_sort_pairs(void* temp, size_t& bytes, const K* keys_in, K* keys_out, const V* vals_in, V* vals_out, size_t n,
                                   unsigned begin_bit = 0, unsigned end_bit = 8 * sizeof(K), hipStream_t = nullptr, bool = false)
{
    if (!temp) { bytes = 256; return hipSuccess; }
    const K mask = (end_bit >= 8 * sizeof(K)) ? ~K(0) : ((K(1) << end_bit) - 1);
    std::vector<size_t> order(n);
    std::iota(order.begin(), order.end(), size_t(0));
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return ((keys_in[a] & mask) >> begin_bit) < ((keys_in[b] & mask) >> begin_bit); });
    for (size_t i = 0; i < n; i++) { keys_out[i] = keys_in[order[i]]; vals_out[i] = vals_in[order[i]]; }
    return hipSuccess;
}
}  // namespace rocprim
