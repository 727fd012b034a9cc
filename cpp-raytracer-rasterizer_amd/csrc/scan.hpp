// scan.hpp -- device-wide exclusive scan of uint32 counts (three small kernels, 1024 items per block);
// used for the rasteriser's rows-per-triangle and the ray tracer's candidates-per-bin tables.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mirt {
constexpr int SCAN_ITEMS = 1024;
__global__ void k_scan_block_sums(const uint32_t *in, int n, uint32_t *sums);
__global__ void k_scan_sums(uint32_t *sums, int nblocks, uint32_t *counters);
__global__ void k_scan_apply(uint32_t *data, int n, const uint32_t *sums, const uint32_t *counters);

// In-place exclusive scan of data[0..n); data[n] and counters[0] receive the total, counters[1] is zeroed.
// `sums` needs n / SCAN_ITEMS + 2 entries.
inline void enqueue_exclusive_scan(uint32_t *data, int n, uint32_t *sums, uint32_t *counters, hipStream_t stream)
{
    const int nblk = (n + SCAN_ITEMS - 1) / SCAN_ITEMS;
    hipLaunchKernelGGL(k_scan_block_sums, dim3(nblk), dim3(256), 0, stream, data, n, sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, stream, sums, nblk, counters);
    hipLaunchKernelGGL(k_scan_apply, dim3(nblk), dim3(256), 0, stream, data, n, sums, counters);
}
}  // namespace mirt
