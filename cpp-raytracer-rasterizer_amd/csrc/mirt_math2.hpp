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

// GLM column-major mat3 times a pair of vectors
__device__ __forceinline__ v3p mat3_mul_vecp(const float *m, v3p v)
{
    return V3P(m[0] * v.x + m[3] * v.y + m[6] * v.z,
               m[1] * v.x + m[4] * v.y + m[7] * v.z,
               m[2] * v.x + m[5] * v.y + m[8] * v.z);
}

}  // namespace mirt
