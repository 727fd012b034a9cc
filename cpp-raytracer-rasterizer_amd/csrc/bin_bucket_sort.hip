// bin_bucket_sort.hip -- orders the (bin, triangle) pairs of k_bin_pairs by bin and produces the per-bin offsets, in two
// launches (round 1 used rocPRIM's generic radix sort + a binary-search kernel: ~22 launches and 75-90 us for the 0.5-1 M
// pairs of a frame, more than the binning itself).
//
// A bin id is at most ~23 bits and the consumers only need "all pairs of a bin are contiguous", so this is a two-level
// counting sort keyed on what the data is.  bucket = bin >> shift, 256 .. 1024 consecutive bins (bucket_sort_shift):
//   k_bin_pairs   (rt_binned.hip) counts the pairs per bucket while it stages them: an LDS histogram per flush, one global
//                 atomic per (flush, non-empty bucket);
//   k_bs_scatter  every workgroup scans the bucket counts for itself (LDS), then takes 4096 pairs at a time, ranks them per
//                 bucket with returning LDS atomics, reserves each bucket's slice with ONE returning global atomic per
//                 (workgroup, bucket) and writes the pairs out -- no per-pair global atomic anywhere (a CU retires one per
//                 ~24 cycles, tools/sortbench.hip);
//   k_bs_local    one workgroup per bucket: counts its pairs per bin in LDS, scans the counts, writes the bins' offsets
//                 (bin_off) and places every triangle id at its final position with LDS cursors; buckets of up to 2048
//                 pairs stay in registers between the two steps.
// The order of the triangle ids INSIDE a bin is not deterministic (LDS atomics); every consumer is order-independent
// (closest hit = minimum of the wavefront min-t key, shadow = any-hit).
#include "bin_sort.hpp"

#include <algorithm>
#include <cstdlib>

namespace mirt {

constexpr int BS_PER_THREAD = 16;                // pairs per thread and chunk of k_bs_scatter (8 / 4: the 100 k soup's binning chain 68 / 75 us against 65)
constexpr int BS_CHUNK = 256 * BS_PER_THREAD;
constexpr int BS_LOCAL_PER_THREAD = 16;
constexpr int BS_MAX_SHIFT = 12;
constexpr int BS_MAX_BUCKET_KEYS = 1 << BS_MAX_SHIFT;    // keys of the largest bucket: one LDS counter each in k_bs_local (16 KiB)

// exclusive scan of v over the 256 threads of the workgroup; *total = sum
__device__ __forceinline__ uint32_t bs_block_scan(uint32_t v, uint32_t *s_wave /* 4 */, uint32_t *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    __syncthreads();                                   // s_wave free again
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t off = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { if (w < wave) off += s_wave[w]; all += s_wave[w]; }
    *total = all;
    return off + incl - v;
}

__global__ __launch_bounds__(256) void k_bs_scatter(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                    const uint32_t *__restrict__ total_ptr, uint32_t cap, uint32_t nbuckets, int shift,
                                                    const uint32_t *__restrict__ bucket_cnt, uint32_t *__restrict__ bucket_base,
                                                    uint32_t *__restrict__ cursor, uint32_t *__restrict__ out_keys, uint32_t *__restrict__ out_vals,
                                                    uint32_t *__restrict__ count_out)
{
    extern __shared__ uint32_t s_dyn[];
    __shared__ uint32_t s_wave[4];
    // the pair count for the host (pinned, mapped memory; nullable): how the next frames size their lists -- stored from here
    // rather than by a copy command, which would sit in the stream between the binning and the sort
    if (count_out && blockIdx.x == 0 && threadIdx.x == 0) *count_out = *total_ptr;
    if (*total_ptr > cap) return;                        // the list overflowed: this frame falls back to brute force (k_rt_trace2), nothing to sort
    uint32_t *s_base = s_dyn, *s_cnt = s_dyn + nbuckets;
    // bucket_base[b] = pairs in buckets < b: every workgroup scans the counts for itself; workgroup 0 publishes the result
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < nbuckets; b0 += 256) {
        const uint32_t b = b0 + threadIdx.x;
        const uint32_t c = b < nbuckets ? bucket_cnt[b] : 0u;
        uint32_t all;
        const uint32_t excl = carry + bs_block_scan(c, s_wave, &all);
        if (b < nbuckets) { s_base[b] = excl; if (blockIdx.x == 0) bucket_base[b] = excl; }
        carry += all;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) bucket_base[nbuckets] = carry;
    const uint32_t total = min(*total_ptr, cap);
    for (uint32_t base = blockIdx.x * BS_CHUNK; base < total; base += gridDim.x * BS_CHUNK) {
        for (uint32_t b = threadIdx.x; b < nbuckets; b += 256) s_cnt[b] = 0u;
        __syncthreads();
        uint32_t k[BS_PER_THREAD], v[BS_PER_THREAD], rank[BS_PER_THREAD];
#pragma unroll
        for (int j = 0; j < BS_PER_THREAD; j++) {
            const uint32_t i = base + threadIdx.x + 256u * j;
            k[j] = 0xFFFFFFFFu; v[j] = 0u; rank[j] = 0u;
            if (i < total) { k[j] = keys[i]; v[j] = vals[i]; }
        }
#pragma unroll
        for (int j = 0; j < BS_PER_THREAD; j++)
            if (k[j] != 0xFFFFFFFFu) rank[j] = atomicAdd(&s_cnt[k[j] >> shift], 1u);
        __syncthreads();
        // where this workgroup's pairs of each bucket go: one returning global atomic per (workgroup, bucket) -- four buckets per
        // thread at a time, their atomics in flight TOGETHER (one after the other, each waiting for its answer, they were most of
        // the ~14 us an iteration took on the 1 M-triangle frame: every workgroup hits every bucket's counter)
        for (uint32_t b0 = threadIdx.x; b0 < nbuckets; b0 += 1024) {
            uint32_t c[4], at[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { const uint32_t b = b0 + 256u * q; c[q] = b < nbuckets ? s_cnt[b] : 0u; at[q] = 0u; }
#pragma unroll
            for (int q = 0; q < 4; q++) if (c[q]) at[q] = atomicAdd(&cursor[b0 + 256u * q], c[q]);
#pragma unroll
            for (int q = 0; q < 4; q++) if (c[q]) s_cnt[b0 + 256u * q] = s_base[b0 + 256u * q] + at[q];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < BS_PER_THREAD; j++)
            if (k[j] != 0xFFFFFFFFu) {
                const uint32_t at = s_cnt[k[j] >> shift] + rank[j];
                out_keys[at] = k[j];
                out_vals[at] = v[j];
            }
        __syncthreads();
    }
}

// One workgroup per bucket of (1 << shift) <= BS_MAX_BUCKET_KEYS keys.  Leaves bucket_cnt[bucket] = cursor[bucket] = 0 for the next sort.
__global__ __launch_bounds__(256) void k_bs_local(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                  const uint32_t *__restrict__ total_ptr, uint32_t cap,
                                                  const uint32_t *__restrict__ bucket_base, uint32_t nbins, int shift,
                                                  uint32_t *__restrict__ bucket_cnt, uint32_t *__restrict__ cursor,
                                                  uint32_t *__restrict__ bin_off, uint32_t *__restrict__ entries)
{
    __shared__ uint32_t s_cnt[BS_MAX_BUCKET_KEYS], s_wave[4];
    const uint32_t bucket = blockIdx.x;
    if (threadIdx.x == 0) { bucket_cnt[bucket] = 0u; cursor[bucket] = 0u; }
    if (*total_ptr > cap) return;                        // overflowed list (see k_bs_scatter): only the counters are reset
    const uint32_t beg = bucket_base[bucket], end = bucket_base[bucket + 1];
    const uint32_t nb = 1u << shift, mask = nb - 1u;
    for (uint32_t i = threadIdx.x; i < nb; i += 256) s_cnt[i] = 0u;
    __syncthreads();
    const bool small = end - beg <= 256u * BS_LOCAL_PER_THREAD;
    uint32_t lo[BS_LOCAL_PER_THREAD], v[BS_LOCAL_PER_THREAD], rank[BS_LOCAL_PER_THREAD];
    if (small) {
#pragma unroll
        for (int j = 0; j < BS_LOCAL_PER_THREAD; j++) {
            const uint32_t i = beg + threadIdx.x + 256u * j;
            lo[j] = 0xFFFFFFFFu; v[j] = 0u; rank[j] = 0u;
            if (i < end) { lo[j] = keys[i] & mask; v[j] = vals[i]; }
        }
#pragma unroll
        for (int j = 0; j < BS_LOCAL_PER_THREAD; j++)
            if (lo[j] != 0xFFFFFFFFu) rank[j] = atomicAdd(&s_cnt[lo[j]], 1u);
    } else {
        for (uint32_t i = beg + threadIdx.x; i < end; i += 256) atomicAdd(&s_cnt[keys[i] & mask], 1u);
    }
    __syncthreads();
    // exclusive scan of the bins' counts -> first entry of every bin
    uint32_t carry = beg;
    for (uint32_t i0 = 0; i0 < nb; i0 += 256) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t c = i < nb ? s_cnt[i] : 0u;
        uint32_t all;
        const uint32_t excl = carry + bs_block_scan(c, s_wave, &all);
        if (i < nb) {
            s_cnt[i] = excl;
            const uint32_t bin = (bucket << shift) + i;
            if (bin <= nbins) bin_off[bin] = excl;           // bin == nbins: the total (that bin holds no pair)
        }
        carry += all;
    }
    __syncthreads();
    if (small) {
#pragma unroll
        for (int j = 0; j < BS_LOCAL_PER_THREAD; j++)
            if (lo[j] != 0xFFFFFFFFu) entries[s_cnt[lo[j]] + rank[j]] = v[j];
    } else {
        for (uint32_t i = beg + threadIdx.x; i < end; i += 256) {
            const uint32_t at = atomicAdd(&s_cnt[keys[i] & mask], 1u);
            entries[at] = vals[i];
        }
    }
}

// keys per bucket = 1 << shift: 256 (measured on the 0.5 M pairs of the 100 k soup's camera frame: scatter + local take 57 /
// 38 / 28 / 25 us at 32 / 64 / 128 / 256 keys per bucket -- fewer buckets mean longer contiguous runs in the scatter's
// output), more -- up to 4096, one LDS counter each in k_bs_local -- where that many buckets would be more than ~1000: every
// flush of k_bin_pairs adds its pairs-per-bucket histogram to the global counts with one atomic per bucket it touched, and the
// pairs of a chunk of a triangle soup touch them all (1 M triangles at 8K, 4.1 M keys: 4050 buckets of 1024 keys made
// k_bin_pairs 2.1 ms of a 1.3 ms frame's GPU time; 1013 buckets of 4096: see DESIGN.md section 5).
int bucket_sort_shift(uint32_t nbins)
{
    int shift = 8;
    while (shift < BS_MAX_SHIFT && ((nbins + 1u + (1u << shift) - 1u) >> shift) > 1024u) shift++;
    // few bins (the 64 x 64 light cube of a moving light: 24 576): smaller buckets, so that k_bs_local has a workgroup per CU
    // and no bucket holds tens of thousands of pairs (one workgroup places a bucket's pairs: 63 -> 20 us on the 100 k soup)
    while (shift > 4 && ((nbins + 1u + (1u << shift) - 1u) >> shift) < 512u) shift--;
    return shift;
}
uint32_t bucket_sort_buckets(uint32_t nbins) { const int s = bucket_sort_shift(nbins); return (nbins + 1u + (1u << s) - 1u) >> s; }

// Enqueues the two launches.  keys/vals: the unsorted pairs (count in *total_ptr, clamped to cap); tmp_keys/tmp_vals: room
// for cap pairs; bucket_cnt (filled by k_bin_pairs; zero on exit), bucket_base (nbuckets + 1), cursor (zero on entry and exit).
hipError_t bucket_sort_pairs(const uint32_t *keys, const uint32_t *vals, const uint32_t *total_ptr, uint32_t cap, uint32_t expected,
                             uint32_t nbins, uint32_t *tmp_keys, uint32_t *tmp_vals, uint32_t *bucket_cnt, uint32_t *bucket_base,
                             uint32_t *cursor, uint32_t *bin_off, uint32_t *entries, int cu_count, hipStream_t stream, uint32_t *count_out)
{
    const int shift = bucket_sort_shift(nbins);
    const uint32_t nbuckets = bucket_sort_buckets(nbins);
    const size_t lds = (size_t)nbuckets * 2 * sizeof(uint32_t);
    uint32_t chunks = (expected + BS_CHUNK - 1) / BS_CHUNK;
    if (chunks < 1) chunks = 1;
    const uint32_t wgs = std::min<uint32_t>(chunks, (uint32_t)cu_count * 2u);
    hipLaunchKernelGGL(k_bs_scatter, dim3(wgs), dim3(256), lds, stream, keys, vals, total_ptr, cap, nbuckets, shift, bucket_cnt, bucket_base,
                       cursor, tmp_keys, tmp_vals, count_out);
    hipLaunchKernelGGL(k_bs_local, dim3(nbuckets), dim3(256), 0, stream, tmp_keys, tmp_vals, total_ptr, cap, bucket_base, nbins, shift, bucket_cnt,
                       cursor, bin_off, entries);
    return hipGetLastError();
}

}  // namespace mirt
