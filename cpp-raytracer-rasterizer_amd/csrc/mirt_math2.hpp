// mirt_math2.hpp -- the arithmetic of mirt_math.hpp on PAIRS of independent values (two pixels per lane).
//
// gfx950 issues one VALU instruction per wave every ~4 cycles whether it is v_mul_f32 or v_pk_mul_f32, and the
// packed forms (v_pk_mul_f32, v_pk_add_f32, v_pk_fma_f32) round each half exactly like their scalar twins
// (measured: profiles/r01_ubench_valu_lds.txt -- 0.24 scalar vs 0.22 packed instructions / clk / SIMD).  So a lane
// that carries two rays does the multiply/add part of its work at twice the rate, bit for bit the same results.
// Every function below keeps the operation order of its scalar original.  Square roots have no packed form and are applied
// per half; a PAIR of divisions shares the multiply-add core of its expansion (div2).
#pragma once

#include "mirt_math.hpp"

namespace mirt {

typedef float f2 __attribute__((ext_vector_type(2)));

struct v3p { f2 x, y, z; };

__device__ __forceinline__ f2 splat2(float a) { return (f2){ a, a }; }
__device__ __forceinline__ v3p V3P(f2 x, f2 y, f2 z) { v3p r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3p splat3(v3 a) { return V3P(splat2(a.x), splat2(a.y), splat2(a.z)); }
__device__ __forceinline__ v3 half0(v3p a) { return V3(a.x.x, a.y.x, a.z.x); }
__device__ __forceinline__ v3 half1(v3p a) { return V3(a.x.y, a.y.y, a.z.y); }
__device__ __forceinline__ v3p join3(v3 a, v3 b) { return V3P((f2){ a.x, b.x }, (f2){ a.y, b.y }, (f2){ a.z, b.z }); }
__device__ __forceinline__ v3p add3p(v3p a, v3p b) { return V3P(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3p sub3p(v3p a, v3p b) { return V3P(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3p mul3p(v3p a, v3p b) { return V3P(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ v3p scale3p(v3p a, f2 s) { return V3P(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3p neg3p(v3p a) { return V3P(-a.x, -a.y, -a.z); }
__device__ __forceinline__ f2 sqrt2(f2 a) { return (f2){ sqrtf(a.x), sqrtf(a.y) }; }

// n / d for both halves, correctly rounded: the compiler's own expansion of an IEEE fp32 division -- v_div_scale x 2, v_rcp,
// six multiply-adds, v_div_fmas, v_div_fixup -- with the six multiply-adds of the two quotients issued as packed instructions
// (v_pk_fma_f32 rounds each half like v_fma_f32).  Same operations on the same operands in the same order as `n.x / d.x` and
// `n.y / d.y`, so the same bits for every input including zeros, infinities, NaNs and subnormals (tools/div2check.hip compares
// 2^28 operand pairs on the device); 16 VALU instructions instead of 22.
__device__ __forceinline__ f2 div2(f2 n, f2 d)
{
    bool unused, c0, c1;
    f2 ds, ns, r;
    ds.x = __builtin_amdgcn_div_scalef(n.x, d.x, false, &unused);
    ds.y = __builtin_amdgcn_div_scalef(n.y, d.y, false, &unused);
    ns.x = __builtin_amdgcn_div_scalef(n.x, d.x, true, &c0);
    ns.y = __builtin_amdgcn_div_scalef(n.y, d.y, true, &c1);
    r.x = __builtin_amdgcn_rcpf(ds.x);
    r.y = __builtin_amdgcn_rcpf(ds.y);
    f2 e = __builtin_elementwise_fma(-ds, r, splat2(1.0f));
    r = __builtin_elementwise_fma(e, r, r);
    f2 q = ns * r;
    e = __builtin_elementwise_fma(-ds, q, ns);
    q = __builtin_elementwise_fma(e, r, q);
    e = __builtin_elementwise_fma(-ds, q, ns);
    return (f2){ __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e.x, r.x, q.x, c0), d.x, n.x),
                 __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e.y, r.y, q.y, c1), d.y, n.y) };
}

// Three quotients over ONE denominator, for both halves: n0/d, n1/d, n2/d (the reference divides a vector by a scalar in
// ClosestIntersection's t, u, v -- raytracer.cpp:237 -- in lightColor / A -- :302, rasteriser.cpp:579 -- and in pPos3d /= zinv,
// rasteriser.cpp:557).  The expansion of an IEEE division refines 1/d before it ever looks at the numerator, so three divisions by
// the same d can share that part -- PROVIDED v_div_scale leaves all operands alone, v_div_fmas is a plain fma and v_div_fixup
// returns its first operand, which is what those instructions do for operands in the middle of the exponent range:
//     2^-40 <= |n_i| < 2^40  and  2^-40 <= |d| < 2^40     (zeros, subnormals, infinities, NaNs are outside)
// (v_div_scale rescales only when an operand or 1/d or n/d is subnormal, |n| < 2^-103, or the exponents differ by >= 96.)  Then the
// instructions below are the very ones `n_i / d` executes, on the same values: rcp, two fma for the reciprocal, and per numerator
// mul, fma, fma, fma, fma -- 19 instructions for the three pairs of quotients instead of 48, two reciprocals instead of six.
// Callers establish the range as cheaply as their operands allow (per span, per light, per triangle -- a per-lane test of all
// eight operands costs more than half of what sharing saves) and pass `outside` = "a half somebody reads may be outside the
// range"; a wave in which any lane says so takes div2 for everything (wave-uniform branch), so the result is the correctly rounded
// quotient in every case.  tools/div3check.hip compares it with `/` on the device: 2^28 random triples inside the range incl.
// its borders, random bit patterns, special values.
constexpr float DIV3_LO = 0x1p-40f, DIV3_HI = 0x1p40f;
// |x| in [2^-40, 2^40): two compares per lane ...
__device__ __forceinline__ bool div3_mag_in_range(float x) { const float a = fabsf(x); return a >= DIV3_LO && a < DIV3_HI; }
// ... or, for values the whole wave shares (kernel arguments), integer arithmetic on the bits, which stays on the scalar unit
__device__ __forceinline__ bool div3_bits_in_range(float x) { return ((__float_as_uint(x) & 0x7fffffffu) - 0x2B800000u) < (0x53800000u - 0x2B800000u); }
__device__ __forceinline__ bool div3_in_range(float a, float b, float c, float d)
{
    const float hi = fmaxf(fmaxf(fabsf(a), fabsf(b)), fabsf(c)), lo = fminf(fminf(fabsf(a), fabsf(b)), fabsf(c));
    // a NaN numerator can hide behind the other two in max3 / min3: the fast path then yields the same quieted NaN as the division
    return hi < DIV3_HI && lo >= DIV3_LO && div3_mag_in_range(d);
}
__device__ __forceinline__ void div3p_fast(f2 n0, f2 n1, f2 n2, f2 d, f2 &q0, f2 &q1, f2 &q2)
{
    f2 r = { __builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y) };
    const f2 e = __builtin_elementwise_fma(-d, r, splat2(1.0f));
    r = __builtin_elementwise_fma(e, r, r);
    f2 a = n0 * r, b = n1 * r, c = n2 * r;
    f2 ea = __builtin_elementwise_fma(-d, a, n0), eb = __builtin_elementwise_fma(-d, b, n1), ec = __builtin_elementwise_fma(-d, c, n2);
    a = __builtin_elementwise_fma(ea, r, a); b = __builtin_elementwise_fma(eb, r, b); c = __builtin_elementwise_fma(ec, r, c);
    ea = __builtin_elementwise_fma(-d, a, n0); eb = __builtin_elementwise_fma(-d, b, n1); ec = __builtin_elementwise_fma(-d, c, n2);
    q0 = __builtin_elementwise_fma(ea, r, a); q1 = __builtin_elementwise_fma(eb, r, b); q2 = __builtin_elementwise_fma(ec, r, c);
}
// `outside`: the wave's lanes (a ballot) that hold a half somebody reads whose operands may lie outside the range (a half nobody
// reads may hold anything and must not send the wave down the general path).
__device__ __forceinline__ void div3p_sel(f2 n0, f2 n1, f2 n2, f2 d, unsigned long long outside, f2 &q0, f2 &q1, f2 &q2)
{
#if defined(MIRT_DIV3_MODE) && MIRT_DIV3_MODE == 0          // (A/B builds: the general division everywhere)
    q0 = div2(n0, d); q1 = div2(n1, d); q2 = div2(n2, d);
#else
    if (__builtin_expect(outside == 0ull, 1)) div3p_fast(n0, n1, n2, d, q0, q1, q2);
    else { q0 = div2(n0, d); q1 = div2(n1, d); q2 = div2(n2, d); }
#endif
}
// The range established per lane from all eight operands.
__device__ __forceinline__ void div3p(f2 n0, f2 n1, f2 n2, f2 d, bool live0, bool live1, f2 &q0, f2 &q1, f2 &q2)
{
    div3p_sel(n0, n1, n2, d, __builtin_amdgcn_ballot_w64((live0 && !div3_in_range(n0.x, n1.x, n2.x, d.x)) || (live1 && !div3_in_range(n0.y, n1.y, n2.y, d.y))), q0, q1, q2);
}
// The same for one value per lane.
__device__ __forceinline__ void div3s_fast(float n0, float n1, float n2, float d, float &q0, float &q1, float &q2)
{
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float a = n0 * r, b = n1 * r, c = n2 * r;
    float ea = __builtin_fmaf(-d, a, n0), eb = __builtin_fmaf(-d, b, n1), ec = __builtin_fmaf(-d, c, n2);
    a = __builtin_fmaf(ea, r, a); b = __builtin_fmaf(eb, r, b); c = __builtin_fmaf(ec, r, c);
    ea = __builtin_fmaf(-d, a, n0); eb = __builtin_fmaf(-d, b, n1); ec = __builtin_fmaf(-d, c, n2);
    q0 = __builtin_fmaf(ea, r, a); q1 = __builtin_fmaf(eb, r, b); q2 = __builtin_fmaf(ec, r, c);
}
__device__ __forceinline__ void div3_sel(float n0, float n1, float n2, float d, unsigned long long outside, float &q0, float &q1, float &q2)
{
    if (__builtin_expect(outside == 0ull, 1)) div3s_fast(n0, n1, n2, d, q0, q1, q2);
    else { q0 = n0 / d; q1 = n1 / d; q2 = n2 / d; }
}
__device__ __forceinline__ void div3(float n0, float n1, float n2, float d, bool live, float &q0, float &q1, float &q2)
{
    div3_sel(n0, n1, n2, d, __builtin_amdgcn_ballot_w64(live && !div3_in_range(n0, n1, n2, d)), q0, q1, q2);
}
// glm::dot: products first, then (x + y) + z
__device__ __forceinline__ f2 dot3p(v3p a, v3p b)
{
    const f2 tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
__device__ __forceinline__ f2 length3p(v3p a) { return sqrt2(dot3p(a, a)); }
__device__ __forceinline__ f2 distance3p(v3p p0, v3p p1) { return length3p(sub3p(p1, p0)); }
// glm::normalize(x) = x * (1 / sqrt(dot(x, x)))
__device__ __forceinline__ v3p normalize3p(v3p a) { return scale3p(a, div2(splat2(1.0f), sqrt2(dot3p(a, a)))); }

// sqrtf for both halves of operands in [2^-96, FLT_MAX]: the compiler's expansion of a correctly rounded square root without the
// part that rescales smaller operands and passes zeros and infinities through -- v_sqrt_f32, then the neighbours one ulp below
// and above are tried against the operand with one fma each -- the two fma of the two halves issued as packed instructions.
// 16 VALU instructions instead of 32 (tools/div3check.hip compares it with sqrtf on the device).
__device__ __forceinline__ f2 sqrt2_fast(f2 x)
{
    const f2 s = { __builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y) };
    const f2 dn = { __uint_as_float(__float_as_uint(s.x) - 1u), __uint_as_float(__float_as_uint(s.y) - 1u) };
    const f2 up = { __uint_as_float(__float_as_uint(s.x) + 1u), __uint_as_float(__float_as_uint(s.y) + 1u) };
    const f2 edn = __builtin_elementwise_fma(-dn, s, x), eup = __builtin_elementwise_fma(-up, s, x);
    f2 r;
    r.x = (0.0f >= edn.x) ? dn.x : s.x; r.x = (0.0f < eup.x) ? up.x : r.x;
    r.y = (0.0f >= edn.y) ? dn.y : s.y; r.y = (0.0f < eup.y) ? up.y : r.y;
    return r;
}
// 1 / d for both halves of operands in [2^-40, 2^40): div3p_fast's steps for the numerator 1 (1 * r is r).
__device__ __forceinline__ f2 rcp2_fast(f2 d)
{
    f2 r = { __builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y) };
    const f2 one = splat2(1.0f);
    f2 e = __builtin_elementwise_fma(-d, r, one);
    r = __builtin_elementwise_fma(e, r, r);
    e = __builtin_elementwise_fma(-d, r, one);
    f2 q = __builtin_elementwise_fma(e, r, r);
    e = __builtin_elementwise_fma(-d, q, one);
    return __builtin_elementwise_fma(e, r, q);
}

// What DirectLight (raytracer.cpp:294-304) and PixelShader (rasteriser.cpp:574-579) compute from a surface point and a light, for
// the two pixels of a lane:  r = glm::distance(pos, L),  rDir = glm::normalize(L - pos),  B = lightColor / (4 pi r^2).
// distance and normalize take the square root of the same dot product; when that dot product d2 lies in [2^-42, 2^36) -- the
// light is between 5e-7 and 2.6e5 away -- r is in [2^-21, 2^18) and A = 4 pi r^2 in [2^-39, 2^40), so the square root needs no
// rescaling and 1 / r and lightColor / A are inside the range of the shared-reciprocal division (`colour_ok`: the light's colour
// is, too -- a property of the frame, tested on the host).  One range test per lane decides for all three; a wave in which any
// live half fails it takes sqrtf and the general divisions, so the results are the correctly rounded ones in every case.
struct LightGeometry2 { f2 r; v3p rDir, B; };
__device__ __forceinline__ LightGeometry2 light_geometry2(v3p pos, v3 L, v3 colour, bool colour_ok, bool live0, bool live1)
{
    LightGeometry2 g;
    const v3p d = sub3p(splat3(L), pos);
    const f2 d2 = dot3p(d, d);
    // [2^-42, 2^36) on the bits of a non-negative float (a negative or NaN d2 is far outside as an unsigned number)
    const bool out0 = (__float_as_uint(d2.x) - 0x2A800000u) >= (0x51800000u - 0x2A800000u);
    const bool out1 = (__float_as_uint(d2.y) - 0x2A800000u) >= (0x51800000u - 0x2A800000u);
    const unsigned long long outside = (__builtin_amdgcn_ballot_w64(out0) & __builtin_amdgcn_ballot_w64(live0)) |
                                       (__builtin_amdgcn_ballot_w64(out1) & __builtin_amdgcn_ballot_w64(live1));
#if defined(MIRT_DIV3_MODE) && MIRT_DIV3_MODE == 0          // (A/B builds: the general operations everywhere)
    if (false) {
#else
    if (__builtin_expect(colour_ok && outside == 0ull, 1)) {
#endif
        g.r = sqrt2_fast(d2);
        g.rDir = scale3p(d, rcp2_fast(g.r));
        const f2 A = { sphere_area(g.r.x), sphere_area(g.r.y) };
        div3p_fast(splat2(colour.x), splat2(colour.y), splat2(colour.z), A, g.B.x, g.B.y, g.B.z);
    } else {
        g.r = sqrt2(d2);
        g.rDir = scale3p(d, div2(splat2(1.0f), g.r));
        const f2 A = { sphere_area(g.r.x), sphere_area(g.r.y) };
        g.B = V3P(div2(splat2(colour.x), A), div2(splat2(colour.y), A), div2(splat2(colour.z), A));
    }
    return g;
}
// GLM column-major mat3 times a pair of vectors
__device__ __forceinline__ v3p mat3_mul_vecp(const float *m, v3p v)
{
    return V3P(m[0] * v.x + m[3] * v.y + m[6] * v.z,
               m[1] * v.x + m[4] * v.y + m[7] * v.z,
               m[2] * v.x + m[5] * v.y + m[8] * v.z);
}

}  // namespace mirt
