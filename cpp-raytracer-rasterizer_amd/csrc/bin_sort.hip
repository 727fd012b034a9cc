// bin_sort.hip -- orders the (bin, triangle) pairs of k_bin_pairs by bin id: rocPRIM's device radix sort over exactly
// the bits a bin id needs.  A plain library primitive in its own translation unit (its headers take seconds to compile
// and need none of the floating-point flags the render kernels are built with).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "bin_sort.hpp"

namespace mirt {

size_t bin_sort_temp_bytes(uint32_t n, int bits)
{
    size_t bytes = 0;
    uint32_t *p = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, bytes, p, p, p, p, (size_t)n, 0u, (unsigned)bits, (hipStream_t)0) != hipSuccess) return 0;
    return bytes;
}

hipError_t bin_sort_pairs(void *temp, size_t temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in,
                          uint32_t *vals_out, uint32_t n, int bits, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, (unsigned)bits, stream);
}

}  // namespace mirt
