// tools/sortbench.hip -- how fast can (bin, triangle) pairs be ordered by bin?  rocPRIM radix sort on the bits a bin id
// needs, against the cost of one global atomic per pair (what the first binner paid twice).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/sortbench.hip -o tools/sortbench ; run on the GPU box.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_atomics(const unsigned *keys, unsigned *cnt, size_t n, int returning, unsigned *out)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (returning) out[i] = atomicAdd(&cnt[keys[i]], 1u);
    else atomicAdd(&cnt[keys[i]], 1u);
}

int main()
{
    for (int cfg = 0; cfg < 2; cfg++) {
        const size_t n = cfg == 0 ? 1800000 : 20000000;
        const unsigned nbins = cfg == 0 ? 57000 : 820000;
        int bits = 1; while ((1u << bits) < nbins) bits++;
        std::vector<unsigned> h(n);
        std::mt19937 rng(1);
        for (auto &x : h) x = rng() % nbins;
        unsigned *k0, *k1, *v0, *v1, *cnt;
        CHECK(hipMalloc(&k0, n * 4)); CHECK(hipMalloc(&k1, n * 4)); CHECK(hipMalloc(&v0, n * 4)); CHECK(hipMalloc(&v1, n * 4)); CHECK(hipMalloc(&cnt, nbins * 4));
        CHECK(hipMemcpy(k0, h.data(), n * 4, hipMemcpyHostToDevice));
        size_t temp = 0;
        CHECK(rocprim::radix_sort_pairs(nullptr, temp, k0, k1, v0, v1, n, 0, bits, (hipStream_t)0));
        void *d_temp; CHECK(hipMalloc(&d_temp, temp));
        hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        float ms;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(a));
            CHECK(rocprim::radix_sort_pairs(d_temp, temp, k0, k1, v0, v1, n, 0, bits, (hipStream_t)0));
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b));
        }
        printf("n=%zu bins=%u bits=%d  radix_sort_pairs %.3f ms (temp %zu B)\n", n, nbins, bits, ms, temp);
        for (int ret = 0; ret < 2; ret++) {
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipMemset(cnt, 0, nbins * 4));
                CHECK(hipEventRecord(a));
                hipLaunchKernelGGL(k_atomics, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, k0, cnt, n, ret, v1);
                CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b));
            }
            printf("   %s atomicAdd per pair: %.3f ms (%.2f G/s)\n", ret ? "returning" : "non-returning", ms, n / ms * 1e-6);
        }
        hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(cnt); hipFree(d_temp);
    }
    return 0;
}
