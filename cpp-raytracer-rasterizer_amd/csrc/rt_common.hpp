// rt_common.hpp -- shared device code of the ray-trace kernels: the origin table, the conservative
// pre-reject filter and the exact accept path of ClosestIntersection (raytracer.cpp:202-257).
#pragma once

#include "mirt_math.hpp"
#include "mirt_math2.hpp"
#include "../../include/mirt.h"

namespace mirt {

// ---- origin table ------------------------------------------------------------------------------------
// For a fixed ray origin S (the camera, or one light) everything in ClosestIntersection that does not
// depend on the ray direction is hoisted into 12 floats per triangle (same operands,
// same operations => same bits as recomputing them per ray):
//   row0 = { e1e2.x, e1e2.y, e1e2.z, e1e2b }    e1e2 = cross(e1,e2), e1e2b = dot(e1e2, b)   (:225,:231)
//   row1 = { be2.x,  be2.y,  be2.z,  near }     be2  = cross(b, e2), b = S - v0              (:218,:226)
//            near = a conservative LOWER bound of the distance from S to any hit point on the triangle (see origin_near)
//   row2 = { e1b.x,  e1b.y,  e1b.z,  0 }        e1b  = cross(e1, b)                          (:227)
// 48 bytes per triangle, read as three 16-byte LDS broadcasts per test.
struct OriginRow { float4 r0, r1, r2; };
static_assert(sizeof(OriginRow) == 48, "origin row must be 48 bytes");

// Rows of the origin table staged into LDS per workgroup pass: 48 KiB at most, 3 workgroups per CU.
constexpr int RT_CHUNK_ROWS = 1024;

// Largest |value| an origin-table entry, a vertex or a light coordinate may have for the pre-reject
// filter's proof to hold (keeps every dot product finite); scenes beyond it take the exact-only path.
#define MIRT_SAFE_MAG 1.0e18f

// A lower bound of glm::distance(S, pos) for every hit point `pos` the reference can compute on this triangle
// (raytracer.cpp:241-242): pos lies in the triangle up to the rounding of v0 + u*e1 + v*e2, so its distance from S is at
// least |c - S| - R for the sphere (c, R) around the vertices; the bound is lowered by 2^-10 of (|c - S| + R) + 1e-6,
// thousands of times the rounding of either side.  Only ever used to SKIP candidates that cannot beat a closer record
// (primary rays) or cannot lie before the shaded point (shadow rays); NaN / Inf never skip (comparisons are `near > x`).
MIRT_HD float origin_near(v3 v0, v3 v1, v3 v2, v3 S)
{
    const v3 c = V3((v0.x + v1.x + v2.x) * (1.0f / 3.0f), (v0.y + v1.y + v2.y) * (1.0f / 3.0f), (v0.z + v1.z + v2.z) * (1.0f / 3.0f));
    const float R = sqrtf(fmaxf(fmaxf(dot3(sub3(v0, c), sub3(v0, c)), dot3(sub3(v1, c), sub3(v1, c))), dot3(sub3(v2, c), sub3(v2, c))));
    const float dc = sqrtf(dot3(sub3(c, S), sub3(c, S)));
    return (dc - R) - (0.0009765625f * (dc + R) + 1.0e-6f);
}

// The matching UPPER bound: no hit point the reference can compute on this triangle lies farther from S (same sphere, same
// slack on the other side).  Lets a shadow ray that surely hits the triangle conclude "occluded" without the exact distance
// (rt_trace.hip, sure_occluder).
MIRT_HD float origin_far(v3 v0, v3 v1, v3 v2, v3 S)
{
    const v3 c = V3((v0.x + v1.x + v2.x) * (1.0f / 3.0f), (v0.y + v1.y + v2.y) * (1.0f / 3.0f), (v0.z + v1.z + v2.z) * (1.0f / 3.0f));
    const float R = sqrtf(fmaxf(fmaxf(dot3(sub3(v0, c), sub3(v0, c)), dot3(sub3(v1, c), sub3(v1, c))), dot3(sub3(v2, c), sub3(v2, c))));
    const float dc = sqrtf(dot3(sub3(c, S), sub3(c, S)));
    return (dc + R) + (0.0009765625f * (dc + R) + 1.0e-6f);
}

MIRT_HD OriginRow make_origin_row(const float *t15, v3 S)
{
    v3 v0 = ld3(t15), v1 = ld3(t15 + 3), v2 = ld3(t15 + 6);
    v3 e1 = sub3(v1, v0), e2 = sub3(v2, v0), b = sub3(S, v0);
    v3 e1e2 = cross3(e1, e2), be2 = cross3(b, e2), e1b = cross3(e1, b);
    float e1e2b = e1e2.x * b.x + e1e2.y * b.y + e1e2.z * b.z;
    OriginRow r;
    r.r0 = make_float4(e1e2.x, e1e2.y, e1e2.z, e1e2b);
    r.r1 = make_float4(be2.x, be2.y, be2.z, origin_near(v0, v1, v2, S));
    r.r2 = make_float4(e1b.x, e1b.y, e1b.z, 0.0f);
    return r;
}

MIRT_HD bool origin_row_safe(const OriginRow &r)
{
    // r0.w (e1e2b) only feeds t = e1e2b / e1e2d and sign tests, so its magnitude is unconstrained
    float m = fmaxf(fmaxf(fmaxf(fabsf(r.r0.x), fabsf(r.r0.y)), fabsf(r.r0.z)),
                    fmaxf(fmaxf(fmaxf(fabsf(r.r1.x), fabsf(r.r1.y)), fabsf(r.r1.z)),
                          fmaxf(fmaxf(fabsf(r.r2.x), fabsf(r.r2.y)), fabsf(r.r2.z))));
    return m < MIRT_SAFE_MAG;   // false for NaN too
}

// ---- the per-test arithmetic ---------------------------------------------------------------------------

struct TestDots { float den, pu, qv; };

// e1e2d, be2d, e1bd (raytracer.cpp:232-234) for negD = nd: three hand-written dot products, products
// first, summed left to right.  Exact reference order; no FMA (the TU is built with -ffp-contract=off).
__device__ __forceinline__ TestDots test_dots(const float4 &r0, const float4 &r1, const float4 &r2, v3 nd)
{
    TestDots d;
    d.den = r0.x * nd.x + r0.y * nd.y + r0.z * nd.z;
    d.pu = r1.x * nd.x + r1.y * nd.y + r1.z * nd.z;
    d.qv = r2.x * nd.x + r2.y * nd.y + r2.z * nd.z;
    return d;
}

// Conservative pre-reject.  Returns false ONLY when the reference's test
//     u + v <= 1 && u >= 0 && v >= 0 && t >= 0,   u = be2d/e1e2d, v = e1bd/e1e2d   (raytracer.cpp:237-239)
// is certain to fail; true means "run the exact path".  With s = sign(e1e2d):  a = s*be2d, b = s*e1bd,
// D = |e1e2d|.  Proof sketch (DESIGN.md section 4.2 has the full argument, valid while every operand is
// finite and below MIRT_SAFE_MAG):
//   u >= 0 (incl. -0 from underflow) needs a > -2^-22, since |e1e2d| < 2^128 keeps |a|/D above 2^-150;
//   likewise v >= 0 needs b > -2^-22;
//   u + v <= 1 after rounding needs a + b <= D*(1 + 2^-22), so fma(D, 1 + 2^-20, -(a+b)) >= 0.
// Anything the filter lets through is decided by the exact divisions, so a loose filter costs time only.
__device__ __forceinline__ bool maybe_hit(const TestDots &d)
{
    const uint32_t sgn = __float_as_uint(d.den) & 0x80000000u;
    const float a = __uint_as_float(__float_as_uint(d.pu) ^ sgn);
    const float b = __uint_as_float(__float_as_uint(d.qv) ^ sgn);
    const float slack = __builtin_fmaf(fabsf(d.den), 1.00000095367431640625f, -(a + b));
    return fminf(fminf(a, b), slack) >= -2.384185791015625e-07f;
}

// A test the reference is CERTAIN to accept (given the filter let it through): with s = sign(e1e2d), a = s*be2d, b = s*e1bd,
// D = |e1e2d|:
//   u = fl(a / D) >= 0 exactly when a >= 0 (a quotient of non-negative operands rounds to a non-negative value; -0 counts as
//   >= 0 on both sides), likewise v and b, and t = fl(e1e2b / e1e2d) with s * e1e2b;
//   fl(u + v) <= 1: u <= (a/D)(1 + 2^-24), v likewise, their rounded sum <= ((a + b)/D)(1 + 2^-24)^2, and fl(a + b) >=
//   (a + b)(1 - 2^-24); so fl(a + b) < fl(D (1 - 2^-20)) gives (a + b)/D < 1 - 2^-21 and the sum stays below 1.  The strict
//   `<` also excludes D == 0 (then a = b = 0 and u = 0/0 is NaN: rejected).  NaN operands fail the comparisons.
// Used by shadow rays only: a certain hit whose every point is closer to the light than 0.99 r (origin_far) occludes -- any-hit is
// exact (SURVEY A-5) -- without the divisions, the hit point or its distance ever being computed.
__device__ __forceinline__ bool sure_hit(const TestDots &d, float e1e2b)
{
    const uint32_t sgn = __float_as_uint(d.den) & 0x80000000u;
    const float a = __uint_as_float(__float_as_uint(d.pu) ^ sgn);
    const float b = __uint_as_float(__float_as_uint(d.qv) ^ sgn);
    const float tn = __uint_as_float(__float_as_uint(e1e2b) ^ sgn);          // s * e1e2b: t = tn / D
    return fminf(a, b) >= 0.0f && (a + b) < fabsf(d.den) * 0.99999904632568359375f && tn >= 0.0f;
}

// ---- the same for two rays per lane (packed FP32, mirt_math2.hpp) ------------------------------------------
struct TestDots2 { f2 den, pu, qv; };

__device__ __forceinline__ TestDots2 test_dots2(const float4 &r0, const float4 &r1, const float4 &r2, const v3p &nd)
{
    TestDots2 d;
    d.den = splat2(r0.x) * nd.x + splat2(r0.y) * nd.y + splat2(r0.z) * nd.z;
    d.pu = splat2(r1.x) * nd.x + splat2(r1.y) * nd.y + splat2(r1.z) * nd.z;
    d.qv = splat2(r2.x) * nd.x + splat2(r2.y) * nd.y + splat2(r2.z) * nd.z;
    return d;
}
__device__ __forceinline__ TestDots dots_half(const TestDots2 &d, int h)
{
    TestDots r;
    r.den = h ? d.den.y : d.den.x; r.pu = h ? d.pu.y : d.pu.x; r.qv = h ? d.qv.y : d.qv.x;
    return r;
}

// maybe_hit (rt_common.hpp) for both rays: a = s*pu, b = s*qv, D = s*den with s = +-1 carrying den's sign --
// multiplying by +-1 is exact, so a, b and D have the very bits the scalar filter's xor / fabs produce.
__device__ __forceinline__ void maybe_hit2(const TestDots2 &d, bool *m0, bool *m1)
{
    const f2 s = { __uint_as_float((__float_as_uint(d.den.x) & 0x80000000u) | 0x3f800000u),
                   __uint_as_float((__float_as_uint(d.den.y) & 0x80000000u) | 0x3f800000u) };
    const f2 a = d.pu * s, b = d.qv * s, D = d.den * s;
    const f2 slack = __builtin_elementwise_fma(D, splat2(1.00000095367431640625f), -(a + b));
    *m0 = fminf(fminf(a.x, b.x), slack.x) >= -2.384185791015625e-07f;
    *m1 = fminf(fminf(a.y, b.y), slack.y) >= -2.384185791015625e-07f;
}

// Whether the three quotients of an accept test (t = e1e2b / e1e2d, u = be2d / e1e2d, v = e1bd / e1e2d, raytracer.cpp:237) may
// leave the range of the shared-reciprocal division (mirt_math2.hpp: div3p_sel).  For a test the filter let through, be2d and
// e1bd are at most |e1e2d| (1 + 2^-19) + 2^-21 in magnitude (maybe_hit: a, b >= -2^-22 and a + b <= D (1 + 2^-20) + 2^-22), so
// |e1e2d| < 2^39 keeps them below 2^40; what is left to test is that none of the three is too small, and e1e2b.
__device__ __forceinline__ bool exact_quotients_outside(float e1e2b, float pu, float qv, float den)
{
    const float lo = fminf(fminf(fabsf(pu), fabsf(qv)), fabsf(den));
    return !(lo >= DIV3_LO && fabsf(den) < 0x1p39f && div3_mag_in_range(e1e2b));
}

// The same filter as the quantity it thresholds: a ray may hit iff its half of the result is >= MAYBE_HIT_THRESHOLD (callers that
// want the verdicts as wave masks compare it themselves).
constexpr float MAYBE_HIT_THRESHOLD = -2.384185791015625e-07f;
__device__ __forceinline__ f2 maybe_hit2_margin(const TestDots2 &d)
{
    const f2 s = { __uint_as_float((__float_as_uint(d.den.x) & 0x80000000u) | 0x3f800000u),
                   __uint_as_float((__float_as_uint(d.den.y) & 0x80000000u) | 0x3f800000u) };
    const f2 a = d.pu * s, b = d.qv * s, D = d.den * s;
    const f2 slack = __builtin_elementwise_fma(D, splat2(1.00000095367431640625f), -(a + b));
    return (f2){ fminf(fminf(a.x, b.x), slack.x), fminf(fminf(a.y, b.y), slack.y) };
}

// The directions of the P rays of a lane: one by one, and -- P == 2 -- as the pairs the packed dot products want, built
// component by component as the directions are computed (pairing the two array elements inside the triangle loop made the
// compiler hoist the pairing through 48 bytes of scratch).
template <int P> struct RayDirs {
    v3 s[P];
    v3p pk;
    __device__ __forceinline__ void set(int p, v3 n)
    {
        s[p] = n;
        if constexpr (P == 2) { pk.x[p] = n.x; pk.y[p] = n.y; pk.z[p] = n.z; }
    }
};

// Dots and filter verdicts of P rays against one origin row: packed when P == 2 and the filter is on.
template <int P, bool FILTER>
__device__ __forceinline__ void test_rays(const float4 &r0, const float4 &r1, const float4 &r2, const RayDirs<P> &rays,
                                          TestDots (&d)[P], bool (&maybe)[P])
{
    const v3 (&nd)[P] = rays.s;
    if constexpr (P == 2 && FILTER) {
        const TestDots2 t = test_dots2(r0, r1, r2, rays.pk);
        maybe_hit2(t, &maybe[0], &maybe[1]);
        d[0] = dots_half(t, 0);
        d[1] = dots_half(t, 1);
    } else {
#pragma unroll
        for (int p = 0; p < P; p++) {
            d[p] = test_dots(r0, r1, r2, nd[p]);
            maybe[p] = !FILTER || maybe_hit(d[p]);
        }
    }
}

// The accept test and hit point exactly as the reference computes them (raytracer.cpp:237-242).
// Returns true when the triangle is accepted; *dist = glm::distance(start, pos).
__device__ __forceinline__ bool exact_hit(const TestDots &d, float e1e2b, const float *t15, v3 start,
                                          v3 *pos, float *dist)
{
    const float t = e1e2b / d.den, u = d.pu / d.den, v = d.qv / d.den;
    if (u + v <= 1.0f && u >= 0.0f && v >= 0.0f && t >= 0.0f) {
        const v3 v0 = ld3(t15), v1 = ld3(t15 + 3), v2 = ld3(t15 + 6);
        const v3 e1 = sub3(v1, v0), e2 = sub3(v2, v0);
        const v3 p = add3(add3(v0, scale3(e1, u)), scale3(e2, v));
        *pos = p;
        *dist = distance3(start, p);
        return true;
    }
    return false;
}

// ---- kernel parameters ---------------------------------------------------------------------------------

struct RtFrame {
    const float *tris15;        // n x 15, reference AoS order
    int n;
    const OriginRow *cam_tab;   // n rows for the camera origin
    const OriginRow *light_tab; // nlights x n rows
    const uint32_t *unsafe;     // != 0: some operand is outside the filter's proven range -> exact-only path
    float cam[3];
    float rot[9];
    float focal;
    int W, H;
    int nlights;                // light POSITIONS to trace shadow rays from: lights x samples (samples consecutive per light)
    int aa;                     // realSamples of Draw(): AA_SAMPLES when AA_ENABLED, else 1 (raytracer.cpp:37-38,549-554)
    int samples;                // soft-shadow samples per light (SOFT_SHADOWS_SAMPLES, raytracer.cpp:41,272-275); 1 = off
    float lpos[MIRT_MAX_LIGHTS][3];
    float lcol[MIRT_MAX_LIGHTS][3];   // P of DirectLight: lights[k].color * lights[k].intensity / samples (raytracer.cpp:282, :296)
    int lights_in_range;        // every lcol component passes light_colour_in_range (mirt_math.hpp)
    float indirect[3];
    int y0, y1, row_origin;
    uint32_t *xrgb;
    int pitch_words;
    float *rgb;                 // nullable, stride W
    int32_t *index;             // nullable, stride W
    float *fd;                  // nullable, stride W: focalDistances = closest distance - FOCAL_LENGTH (raytracer.cpp:248-249), 0 on a miss
    float focal_plane;          // FOCAL_LENGTH (:45)
    float *dist;                // nullable, stride W: closestIntersections[].distance (raytracer.cpp:91-98), FLT_MAX on a miss (:335-339)
    float *pos;                 // nullable, stride W, 3 floats per pixel: closestIntersections[].position, 0 on a miss
    unsigned long long *hit_count;   // HIT_SHARDS counters, HIT_SHARD_STRIDE u64 apart (see count_hits)
};

// Every wave reports how many of its primary rays hit (the shadow-ray count of the frame).  One counter for the
// whole grid serialises ~30k atomics on one address (measured: ~7 ns each, i.e. the entire kernel time of a
// Cornell-box frame), so the count is spread over HIT_SHARDS addresses on separate 128-byte lines and summed
// on the host in mirt_get_stats().
constexpr int HIT_SHARDS = 256;
constexpr int HIT_SHARD_STRIDE = 16;

__device__ __forceinline__ void count_hits(const RtFrame &f, unsigned long long wave_hits)
{
    if ((threadIdx.x & 63) == 0 && wave_hits) {
        const unsigned shard = (blockIdx.y * gridDim.x + blockIdx.x + (threadIdx.x >> 6) * 61u) % HIT_SHARDS;
        atomicAdd(f.hit_count + (size_t)shard * HIT_SHARD_STRIDE, wave_hits);
    }
}

// 64-lane sum (DPP/ds_swizzle-free shuffle tree); every lane returns the total
__device__ __forceinline__ unsigned wave_sum(unsigned v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// Sub-ray stepping of Draw()'s supersampling loops (raytracer.cpp:566-596): y1/x1 start at the pixel centre, or half a
// pixel before it when realSamples > 1, and advance by 1/(realSamples-1).  (With realSamples == 1 that increment is
// 1/0 = +inf, exactly as in the reference; the value is never used again.)
__device__ __forceinline__ float aa_start(int c, int rs) { return rs > 1 ? (float)c - 0.5f : (float)c; }
__device__ __forceinline__ float aa_step(int rs) { return 1.0f / (float)(rs - 1); }

// 64-lane min / max without LDS traffic: four row_shr steps fold each row of 16 lanes into its lane 15, row_bcast:15
// and row_bcast:31 carry the row results into lane 63, v_readlane makes the result wave-uniform (an SGPR).  Lanes a
// step has no source for are skipped by the hardware and keep their value.  Written as v_min_f32_dpp / v_max_f32_dpp
// directly (the compiler keeps a separate v_mov_b32_dpp per step otherwise); the s_nop 1 are the two wait states a
// DPP read needs after a VALU write of the same register.
#define MIRT_DPP_REDUCE(OP)                                                                      \
    asm("s_nop 1\n\t" OP " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                \
        "s_nop 1\n\t" OP " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"                \
        "s_nop 1\n\t" OP " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"                \
        "s_nop 1\n\t" OP " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"                \
        "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"             \
        "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"             \
        "s_nop 1" : "+v"(v))
__device__ __forceinline__ float wave_min_f(float v)
{
    MIRT_DPP_REDUCE("v_min_f32_dpp");
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_f(float v)
{
    MIRT_DPP_REDUCE("v_max_f32_dpp");
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#undef MIRT_DPP_REDUCE

// ---- wavefront min-t primitive ------------------------------------------------------------------------
// The reference's sequential closest-hit update, `if (closest.distance >= distance)` in index order
// (raytracer.cpp:243-247), keeps the smallest distance and, among exact ties, the LARGEST index.  Distances are
// >= 0, so their float bits order as unsigned integers; packing  key = dist_bits << 32 | (0xFFFFFFFF - index)
// turns that rule into a plain unsigned minimum, which 64 lanes reduce with a 6-step xor butterfly.
__device__ __forceinline__ unsigned long long min_t_key(float dist, int index)
{
    return ((unsigned long long)__float_as_uint(dist) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)index);
}
__device__ __forceinline__ int min_t_index(unsigned long long key) { return (int)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull)); }
__device__ __forceinline__ float min_t_dist(unsigned long long key) { return __uint_as_float((uint32_t)(key >> 32)); }
constexpr unsigned long long MIN_T_NONE = ((unsigned long long)0x7F7FFFFFu << 32) | 0xFFFFFFFFull;   // FLT_MAX, "index -1"

// DPP reduction (no LDS traffic, unlike a __shfl_xor butterfly, which is ds_bpermute): four row_shr steps fold each row of 16
// lanes into its lane 15, row_bcast:15 / row_bcast:31 carry the row results into lane 63 (lanes a step has no source for keep
// their own value), v_readlane makes the minimum wave-uniform.  The two halves of the 64-bit key move in lockstep.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_min_step(unsigned long long key)
{
    const int lo = (int)(uint32_t)key, hi = (int)(uint32_t)(key >> 32);
    const uint32_t olo = (uint32_t)__builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const uint32_t ohi = (uint32_t)__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
    return o < key ? o : key;
}

__device__ __forceinline__ unsigned long long wave_min_key(unsigned long long key)
{
    key = dpp_min_step<0x111, 0xf>(key);      // row_shr:1
    key = dpp_min_step<0x112, 0xf>(key);      // row_shr:2
    key = dpp_min_step<0x114, 0xf>(key);      // row_shr:4
    key = dpp_min_step<0x118, 0xf>(key);      // row_shr:8
    key = dpp_min_step<0x142, 0xa>(key);      // row_bcast:15 into rows 1 and 3
    key = dpp_min_step<0x143, 0xc>(key);      // row_bcast:31 into rows 2 and 3
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, 63);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;       // every lane holds the wave minimum
}

// ray-triangle tests a wave executed (second word of the shard); used for the roofline of the binned path
__device__ __forceinline__ void count_tests(const RtFrame &f, unsigned lane_tests)
{
    const unsigned total = wave_sum(lane_tests);
    if ((threadIdx.x & 63) == 0 && total) {
        const unsigned shard = (blockIdx.y * gridDim.x + blockIdx.x + (threadIdx.x >> 6) * 61u) % HIT_SHARDS;
        atomicAdd(f.hit_count + (size_t)shard * HIT_SHARD_STRIDE + 1, (unsigned long long)total);
    }
}

// closestIntersections[pixel] as the frame leaves it (struct Intersection, raytracer.cpp:91-98): distance is the FLT_MAX of
// Update()'s reset (:335-339) where no triangle was accepted; the reference leaves position uninitialised there, here it is 0.
__device__ __forceinline__ void store_intersection(const RtFrame &f, size_t px, int best_i, float best_d, v3 pos)
{
    if (f.dist) f.dist[px] = best_i >= 0 ? best_d : 3.402823466e+38f;
    if (f.pos) st3(f.pos + 3 * px, best_i >= 0 ? pos : V3(0.0f, 0.0f, 0.0f));
}

// candidates a wave OFFERED to its rays (third word of the shard): list entries x rays, whether the filter ran on them or a
// `near` bound skipped them -- what round 1's kernels counted as tests, kept so that the two rounds compare like for like
__device__ __forceinline__ void count_candidates(const RtFrame &f, unsigned lane_candidates)
{
    const unsigned total = wave_sum(lane_candidates);
    if ((threadIdx.x & 63) == 0 && total) {
        const unsigned shard = (blockIdx.y * gridDim.x + blockIdx.x + (threadIdx.x >> 6) * 61u) % HIT_SHARDS;
        atomicAdd(f.hit_count + (size_t)shard * HIT_SHARD_STRIDE + 2, (unsigned long long)total);
    }
}

// Per-light shading term D of DirectLight (raytracer.cpp:294-304) before the shadow test.
__device__ __forceinline__ v3 light_term(const RtFrame &f, int k, v3 hit, v3 nDir, v3 *rDir, float *r)
{
    const v3 L = ld3(f.lpos[k]);
    const float rr = distance3(hit, L);
    const float A = sphere_area(rr);
    const v3 P = ld3(f.lcol[k]);                       // P = lightColor /= (float)samples (:296), divided on the host
    const v3 rd = normalize3(sub3(L, hit));
    const v3 B = div3s(P, A);
    const float d = dot3(rd, nDir);
    const float m = (d < 0.0f) ? 0.0f : d;             // std::max(d, 0.0f)
    *rDir = rd;
    *r = rr;
    return scale3(B, m);
}

}  // namespace mirt
