// blend_sort.hip -- device radix sort of the transparent pass' fragment list (row N3): 64-bit keys
// (pixel sample << 32 | draw order) with 32-bit payloads, through hipCUB (rocPRIM).  A separate translation unit so
// that the library's main file does not pay for the hipCUB headers.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

extern "C" int r3n_internal_sort_pairs(void *temp, size_t *temp_bytes, const unsigned long long *keys_in,
                                       unsigned long long *keys_out, const unsigned int *vals_in, unsigned int *vals_out,
                                       unsigned int n, int end_bit, hipStream_t stream) {
    return (int)hipcub::DeviceRadixSort::SortPairs(temp, *temp_bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, end_bit,
                                                   stream);
}
