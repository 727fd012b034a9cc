// tools/div3check.hip -- (GPU box) div3p / div3 of csrc/mirt_math2.hpp against the compiler's own `/` on the device:
//   mode 0  2^28 triples over one denominator with uniform random mantissas and signs and exponents uniform over the fast path's
//           range [2^-40, 2^40) -- every lane takes the shared-reciprocal path;
//   mode 1  uniform random bit patterns (every exponent incl. subnormals, infinities, NaNs): the general path;
//   mode 2  triples built from special values (zeros, subnormals, infinities, NaN, the range's border values and their neighbours);
//   mode 3  in-range waves with ONE lane holding a special value: the wave-uniform branch must send everybody down the general path.
// Every mode also runs sqrt2_fast / rcp2_fast (mode 0) and light_geometry2 -- distance, normalised direction and lightColor / (4 pi r^2)
// for points at ordinary distances from the light and at distances of every exponent -- against the plain expressions.
// Bits are compared (any NaN equals any NaN).  Prints the mismatch counts; exit code 1 if there was one.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "../cpp-raytracer-rasterizer_amd/csrc/mirt_math2.hpp"
using namespace mirt;

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ bool same(float a, float b) { return (a != a && b != b) || __float_as_uint(a) == __float_as_uint(b); }
// a value with a random sign and mantissa and a biased exponent uniform over [87, 167) = [2^-40, 2^40)
__device__ __forceinline__ uint32_t in_range_bits(uint32_t h) { return (h & 0x807fffffu) | ((87u + ((h >> 23) & 0xffu) % 80u) << 23); }

__global__ void k_check(unsigned long long *bad, uint32_t seed, int mode)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t sp[16] = { 0u, 0x80000000u, 1u, 0x007fffffu, 0x00800000u, 0x7f7fffffu, 0x7f800000u, 0xff800000u, 0x7fc00000u,
                              0x3f800000u, 0xbf800000u, 0x53800000u, 0x2b800000u, 0x537fffffu, 0x2b7fffffu, 0x3f7fffffu };   // incl. 2^40, 2^-40 and their predecessors
    uint32_t w[8];
    for (int k = 0; k < 8; k++) w[k] = mix(i * 8u + (uint32_t)k + seed);
    if (mode == 0 || mode == 3) for (int k = 0; k < 8; k++) w[k] = in_range_bits(w[k]);
    if (mode == 2) {
        w[0] = sp[i & 15]; w[3] = sp[(i >> 4) & 15];
        if (i & 256) w[1] = sp[(i >> 9) & 15];
        if (i & 8192) w[2] = sp[(i >> 14) & 15];
        w[4] = in_range_bits(w[4]); w[5] = in_range_bits(w[5]); w[6] = in_range_bits(w[6]);
        w[7] = (i & (1u << 18)) ? in_range_bits(w[7]) : sp[(i >> 19) & 15];
    }
    if (mode == 3 && (i & 63) == (seed & 63)) w[(i >> 6) & 7] = sp[(i >> 9) & 15];
    const float a0 = __uint_as_float(w[0]), a1 = __uint_as_float(w[1]), a2 = __uint_as_float(w[2]), ad = __uint_as_float(w[3]);
    const float b0 = __uint_as_float(w[4]), b1 = __uint_as_float(w[5]), b2 = __uint_as_float(w[6]), bd = __uint_as_float(w[7]);
    f2 q0, q1, q2;
    div3p((f2){ a0, b0 }, (f2){ a1, b1 }, (f2){ a2, b2 }, (f2){ ad, bd }, true, true, q0, q1, q2);
    int wrong = !same(q0.x, a0 / ad) + !same(q1.x, a1 / ad) + !same(q2.x, a2 / ad) + !same(q0.y, b0 / bd) + !same(q1.y, b1 / bd) + !same(q2.y, b2 / bd);
    float s0, s1, s2;
    div3(a0, a1, a2, ad, true, s0, s1, s2);
    wrong += !same(s0, a0 / ad) + !same(s1, a1 / ad) + !same(s2, a2 / ad);
    // a half nobody reads must not matter: garbage in the dead half, the live half still right
    div3p((f2){ a0, __uint_as_float(0x7fc00000u) }, (f2){ a1, 0.0f }, (f2){ a2, b2 }, (f2){ ad, 0.0f }, true, false, q0, q1, q2);
    wrong += !same(q0.x, a0 / ad) + !same(q1.x, a1 / ad) + !same(q2.x, a2 / ad);
    if (mode == 0) {
        // sqrt2_fast against sqrtf on [2^-96, FLT_MAX] (exponents spread over the whole range), rcp2_fast against 1 / d in range
        const float x0 = __uint_as_float((w[0] & 0x007fffffu) | ((31u + ((w[1] >> 3) % 224u)) << 23)), x1 = __uint_as_float((w[4] & 0x007fffffu) | ((31u + ((w[5] >> 3) % 224u)) << 23));
        const f2 sq = sqrt2_fast((f2){ x0, x1 });
        wrong += !same(sq.x, sqrtf(x0)) + !same(sq.y, sqrtf(x1));
        const f2 rc = rcp2_fast((f2){ ad, bd });
        wrong += !same(rc.x, 1.0f / ad) + !same(rc.y, 1.0f / bd);
    }
    {
        // light_geometry2 against the plain expressions: points whose distance from the light spreads over every exponent in modes
        // 1 and 2 (both sides of its range test), ordinary distances in modes 0 and 3
        const float sc = (mode == 0 || mode == 3) ? 1.0f : __uint_as_float(((w[6] >> 2) % 254u + 1u) << 23);
        const v3 L = V3(0.1f, -0.5f, -0.7f), col = V3(14.0f, mode == 2 && (i & 1) ? 0.0f : 7.0f, 3.5f);
        const v3 pa = V3(L.x + sc * (a0 / fmaxf(fabsf(a0), 1e-30f)) * 0.37f, L.y + sc * __uint_as_float((w[1] & 0x007fffffu) | 0x3f000000u), L.z - sc * __uint_as_float((w[2] & 0x007fffffu) | 0x3e800000u));
        const v3 pb = V3(L.x - sc * 0.61f, L.y + sc * __uint_as_float((w[5] & 0x007fffffu) | 0x3f800000u), L.z + sc * __uint_as_float((w[6] & 0x007fffffu) | 0x3f000000u));
        const LightGeometry2 g = light_geometry2(join3(pa, pb), L, col, light_colour_in_range(&col.x), true, (i & 64) != 0);
        const v3 da = sub3(L, pa), db = sub3(L, pb);
        const float ra = sqrtf(dot3(da, da)), rb = sqrtf(dot3(db, db));
        const v3 na = scale3(da, 1.0f / ra), nb = scale3(db, 1.0f / rb);
        const float Aa = sphere_area(ra), Ab = sphere_area(rb);
        wrong += !same(g.r.x, ra) + !same(g.rDir.x.x, na.x) + !same(g.rDir.y.x, na.y) + !same(g.rDir.z.x, na.z) + !same(g.B.x.x, col.x / Aa) + !same(g.B.y.x, col.y / Aa) + !same(g.B.z.x, col.z / Aa);
        if (i & 64) wrong += !same(g.r.y, rb) + !same(g.rDir.x.y, nb.x) + !same(g.rDir.y.y, nb.y) + !same(g.rDir.z.y, nb.z) + !same(g.B.x.y, col.x / Ab) + !same(g.B.y.y, col.y / Ab) + !same(g.B.z.y, col.z / Ab);
    }
    if (wrong) atomicAdd(bad + mode, (unsigned long long)wrong);
    if (mode == 0 && !(div3_in_range(a0, a1, a2, ad) && div3_in_range(b0, b1, b2, bd))) atomicAdd(bad + 4, 1ull);
}

int main()
{
    unsigned long long *bad, h[5] = { 0, 0, 0, 0, 0 };
    hipMalloc(&bad, 40); hipMemset(bad, 0, 40);
    for (int r = 0; r < 64; r++) hipLaunchKernelGGL(k_check, dim3(1 << 13), dim3(256), 0, 0, bad, 0x9e3779b9u * (uint32_t)(r + 1), 0);   // 64 x 2^21 lanes x 2 halves = 2^28 triples
    for (int r = 0; r < 16; r++) hipLaunchKernelGGL(k_check, dim3(1 << 13), dim3(256), 0, 0, bad, 0x85ebca6bu * (uint32_t)(r + 1), 1);
    for (int r = 0; r < 4; r++) hipLaunchKernelGGL(k_check, dim3(1 << 12), dim3(256), 0, 0, bad, 77u + (uint32_t)r, 2);
    for (int r = 0; r < 64; r++) hipLaunchKernelGGL(k_check, dim3(1 << 10), dim3(256), 0, 0, bad, 1000u + (uint32_t)r, 3);
    hipDeviceSynchronize();
    hipMemcpy(h, bad, 40, hipMemcpyDeviceToHost);
    printf("div3 vs '/': mismatching quotients -- in range (2^28 triples): %llu, random bits: %llu, special values: %llu, one odd lane per wave: %llu; "
           "in-range operands the range test refused: %llu\n", h[0], h[1], h[2], h[3], h[4]);
    return (h[0] | h[1] | h[2] | h[3] | h[4]) ? 1 : 0;
}
