// bin_sort.hpp -- radix sort of (bin, triangle) pairs by bin id (bin_sort.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace mirt {

// temporary storage the sort of n pairs on `bits` key bits needs (0 on failure)
size_t bin_sort_temp_bytes(uint32_t n, int bits);
// keys_out / vals_out receive the pairs ordered by the low `bits` bits of the key
hipError_t bin_sort_pairs(void *temp, size_t temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in,
                          uint32_t *vals_out, uint32_t n, int bits, hipStream_t stream);

}  // namespace mirt
