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

// bin_bucket_sort.hip: two-level counting sort on the bin id (bucket = bin >> shift) that also produces bin_off; see there.
int bucket_sort_shift(uint32_t nbins);
uint32_t bucket_sort_buckets(uint32_t nbins);
constexpr uint32_t BUCKET_SORT_MAX_BUCKETS = 8192;       // one LDS counter per bucket (32 KiB in k_bin_pairs, 64 KiB in k_bs_scatter)
hipError_t bucket_sort_pairs(const uint32_t *keys, const uint32_t *vals, const uint32_t *total_ptr, uint32_t cap, uint32_t expected,
                             uint32_t nbins, uint32_t *tmp_keys, uint32_t *tmp_vals, uint32_t *bucket_cnt, uint32_t *bucket_base,
                             uint32_t *cursor, uint32_t *bin_off, uint32_t *entries, int cu_count, hipStream_t stream, uint32_t *count_out = nullptr);

}  // namespace mirt
