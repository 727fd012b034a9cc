// bin_sort.hpp -- ordering the (bin, triangle) pairs of a binning pass by key (bin_bucket_sort.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace mirt {

// bin_bucket_sort.hip: two-level counting sort on the bin id (bucket = bin >> shift) that also produces bin_off; see there.
int bucket_sort_shift(uint32_t nbins);
uint32_t bucket_sort_buckets(uint32_t nbins);
constexpr uint32_t BUCKET_SORT_MAX_BUCKETS = 8192;       // one LDS counter per bucket (32 KiB in k_bin_pairs, 64 KiB in k_bs_scatter); buckets hold 256 .. 4096 keys
hipError_t bucket_sort_pairs(const uint32_t *keys, const uint32_t *vals, const uint32_t *total_ptr, uint32_t cap, uint32_t expected,
                             uint32_t nbins, uint32_t *tmp_keys, uint32_t *tmp_vals, uint32_t *bucket_cnt, uint32_t *bucket_base,
                             uint32_t *cursor, uint32_t *bin_off, uint32_t *entries, int cu_count, hipStream_t stream, uint32_t *count_out = nullptr);

}  // namespace mirt
