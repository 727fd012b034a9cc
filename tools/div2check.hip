// tools/div2check.hip -- (GPU box) div2 of csrc/mirt_math2.hpp against the compiler's own `/` on the device: 2^28 operand pairs
// (uniform random bit patterns -- every exponent incl. subnormals, infinities, NaNs -- plus pairs built from special values),
// compared as bits (any NaN equals any NaN).  Prints the mismatch count; exit code 1 if there was one.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "../cpp-raytracer-rasterizer_amd/csrc/mirt_math2.hpp"
using namespace mirt;

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ bool same(float a, float b) { return (a != a && b != b) || __float_as_uint(a) == __float_as_uint(b); }

__global__ void k_check(unsigned long long *bad, uint32_t seed, int special)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t sp[16] = { 0u, 0x80000000u, 1u, 0x007fffffu, 0x00800000u, 0x7f7fffffu, 0x7f800000u, 0xff800000u, 0x7fc00000u,
                              0x3f800000u, 0xbf800000u, 0x5f800000u, 0x1f800000u, 0x7e800000u, 0x00ffffffu, 0x3f7fffffu };
    uint32_t a = mix(i * 4u + seed), b = mix(i * 4u + 1u + seed), c = mix(i * 4u + 2u + seed), d = mix(i * 4u + 3u + seed);
    if (special) { a = sp[i & 15]; b = (i & 16) ? sp[(i >> 5) & 15] : b; c = (i & 512) ? sp[(i >> 10) & 15] : c; d = sp[(i >> 14) & 15]; }
    const f2 n = { __uint_as_float(a), __uint_as_float(c) }, dd = { __uint_as_float(b), __uint_as_float(d) };
    const f2 q = div2(n, dd);
    const float q0 = n.x / dd.x, q1 = n.y / dd.y;
    if (!same(q.x, q0) || !same(q.y, q1)) atomicAdd(bad, 1ull);
}

int main()
{
    unsigned long long *bad, h = 0;
    hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
    for (int r = 0; r < 64; r++) hipLaunchKernelGGL(k_check, dim3(1 << 13), dim3(256), 0, 0, bad, 0x9e3779b9u * (uint32_t)(r + 1), 0);   // 64 x 2^21 lanes x 2 halves = 2^28
    for (int r = 0; r < 4; r++) hipLaunchKernelGGL(k_check, dim3(1 << 12), dim3(256), 0, 0, bad, 77u + (uint32_t)r, 1);
    hipDeviceSynchronize();
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("div2 vs '/': %llu mismatching lanes of %llu\n", h, (unsigned long long)(64ull * (1ull << 21) + 4ull * (1ull << 20)));
    return h ? 1 : 0;
}
