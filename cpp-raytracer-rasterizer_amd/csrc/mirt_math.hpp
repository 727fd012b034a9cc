// mirt_math.hpp -- the reference's float arithmetic, operation for operation, usable on host and device.
//
// Every function here reproduces the evaluation order of the GLM 0.9.7.2 routine the reference calls
// (raytracer/glm/detail/func_geometric.inl:65-72,94-115,133-159; type_mat3x3.inl:36-56,506-522;
// type_vec3.inl:300-308,703-709; func_common.inl:409-456), because "closest-hit triangle index bit-exact"
// needs bit-exact distances: Cornell-box quads share diagonals, so exact ties decide 1210 of 250000 pixels.
// The translation unit MUST be compiled with -ffp-contract=off (no FMA formation) and without fast-math;
// hipcc's default correctly-rounded fp32 divide / sqrt and preserved subnormals are relied upon.
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define MIRT_HD __host__ __device__ __forceinline__

namespace mirt {

struct v3 { float x, y, z; };

MIRT_HD v3 V3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
MIRT_HD v3 ld3(const float *p) { return V3(p[0], p[1], p[2]); }
MIRT_HD void st3(float *p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
MIRT_HD v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
MIRT_HD v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
MIRT_HD v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
MIRT_HD v3 scale3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
MIRT_HD v3 div3s(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
MIRT_HD v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }

// glm::dot: products first, then (x + y) + z
MIRT_HD float dot3(v3 a, v3 b)
{
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
// glm::cross
MIRT_HD v3 cross3(v3 x, v3 y)
{
    return V3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
MIRT_HD float length3(v3 a) { return sqrtf(dot3(a, a)); }
// glm::distance(p0, p1) = length(p1 - p0)
MIRT_HD float distance3(v3 p0, v3 p1) { return length3(sub3(p1, p0)); }
// glm::normalize(x) = x * (1 / sqrt(dot(x, x)))
MIRT_HD v3 normalize3(v3 a) { return scale3(a, 1.0f / sqrtf(dot3(a, a))); }

// GLM column-major mat3: m[c*3+r] == m[c][r]
MIRT_HD v3 mat3_mul_vec(const float *m, v3 v)
{
    return V3(m[0] * v.x + m[3] * v.y + m[6] * v.z,
              m[1] * v.x + m[4] * v.y + m[7] * v.z,
              m[2] * v.x + m[5] * v.y + m[8] * v.z);
}
MIRT_HD v3 vec_mul_mat3(v3 v, const float *m)
{
    return V3(m[0] * v.x + m[1] * v.y + m[2] * v.z,
              m[3] * v.x + m[4] * v.y + m[5] * v.z,
              m[6] * v.x + m[7] * v.y + m[8] * v.z);
}
// glm::inverse(mat3): cofactors times OneOverDeterminant
MIRT_HD void mat3_inverse(const float *m, float *inv)
{
#define MIRT_M(c, r) m[(c) * 3 + (r)]
    float ood = 1.0f / (+MIRT_M(0, 0) * (MIRT_M(1, 1) * MIRT_M(2, 2) - MIRT_M(2, 1) * MIRT_M(1, 2))
                        - MIRT_M(1, 0) * (MIRT_M(0, 1) * MIRT_M(2, 2) - MIRT_M(2, 1) * MIRT_M(0, 2))
                        + MIRT_M(2, 0) * (MIRT_M(0, 1) * MIRT_M(1, 2) - MIRT_M(1, 1) * MIRT_M(0, 2)));
    inv[0] = +(MIRT_M(1, 1) * MIRT_M(2, 2) - MIRT_M(2, 1) * MIRT_M(1, 2)) * ood;
    inv[3] = -(MIRT_M(1, 0) * MIRT_M(2, 2) - MIRT_M(2, 0) * MIRT_M(1, 2)) * ood;
    inv[6] = +(MIRT_M(1, 0) * MIRT_M(2, 1) - MIRT_M(2, 0) * MIRT_M(1, 1)) * ood;
    inv[1] = -(MIRT_M(0, 1) * MIRT_M(2, 2) - MIRT_M(2, 1) * MIRT_M(0, 2)) * ood;
    inv[4] = +(MIRT_M(0, 0) * MIRT_M(2, 2) - MIRT_M(2, 0) * MIRT_M(0, 2)) * ood;
    inv[7] = -(MIRT_M(0, 0) * MIRT_M(2, 1) - MIRT_M(2, 0) * MIRT_M(0, 1)) * ood;
    inv[2] = +(MIRT_M(0, 1) * MIRT_M(1, 2) - MIRT_M(1, 1) * MIRT_M(0, 2)) * ood;
    inv[5] = -(MIRT_M(0, 0) * MIRT_M(1, 2) - MIRT_M(1, 0) * MIRT_M(0, 2)) * ood;
    inv[8] = +(MIRT_M(0, 0) * MIRT_M(1, 1) - MIRT_M(1, 0) * MIRT_M(0, 1)) * ood;
#undef MIRT_M
}

// float -> int the way the reference's x86-64 build converts (cvttss2si): truncate; NaN / out of range
// give INT_MIN.  (The GPU's v_cvt_i32_f32 saturates instead, so the range check is explicit.)
MIRT_HD int f2i_x86(float f)
{
    if (!(f > -2147483904.0f && f < 2147483648.0f)) return (int)0x80000000;
    return (int)f;
}

// PutPixelSDL's conversion (raytracer/Source/SDLauxiliary.h:75-80): Uint8(clamp(255*c, 0, 255)) per
// channel, packed r<<16 | g<<8 | b as SDL_MapRGB does on the 32-bit XRGB8888 software surface.
MIRT_HD uint32_t chan8(float c)
{
    float v = 255 * c;
    // glm::clamp = min(max(x, 0), 255) with glm's `(x < y) ? y : x` forms: NaN becomes 0.  fmaxf / fminf return the
    // non-NaN operand, so they give the same value (up to the sign of a zero, which the conversion drops) in one
    // instruction each instead of a compare and a select.
    v = fmaxf(v, 0.0f);
    v = fminf(v, 255.0f);
    return (uint32_t)(int)v;
}
MIRT_HD uint32_t pack_xrgb(v3 c) { return (chan8(c.x) << 16) | (chan8(c.y) << 8) | chan8(c.z); }

// A = 4*M_PI*(r*r): the product r*r is float, the rest double, narrowed on assignment
// (raytracer.cpp:295, rasteriser.cpp:575).
MIRT_HD float sphere_area(float r) { return (float)(4 * 3.14159265358979323846 * (double)(r * r)); }

// Whether a light's colour (times its intensity, over the shadow samples) is inside [2^-40, 2^40) in every component: the range in
// which the kernels may share one refined reciprocal between the three divisions lightColor / A (mirt_math2.hpp: light_geometry2).
MIRT_HD bool light_colour_in_range(const float *c)
{
    for (int k = 0; k < 3; k++) { const float a = c[k] < 0.0f ? -c[k] : c[k]; if (!(a >= 0x1p-40f && a < 0x1p40f)) return false; }
    return true;
}

}  // namespace mirt
