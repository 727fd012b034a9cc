/*
 * oracle/ref_model.cpp -- TEST INFRASTRUCTURE ONLY (never linked into libmirt.so).
 *
 * The part of the real reference that compiles from its own files in this image: the scene model
 * header raytracer/Source/TestModel.h (Triangle, Light, LoadTestModel) and the GLM 0.9.7.2 headers the
 * reference vendors under raytracer/glm.  Both are #included where they lie under /root/reference
 * (found through -I, see oracle/Makefile); nothing is copied, the built library goes to oracle/_ref/
 * (git-ignored).  The render translation units themselves need <SDL.h> (absent here, no stand-in
 * allowed) and are therefore NOT built.
 *
 * Used by tests/test_oracle_ref_model.py to check, bit for bit, that oracle/mirt_oracle.c restates
 * (a) LoadTestModel and (b) the operation order of every GLM function on the render path.
 */
#include <cstdint>
#include <vector>
#include <glm/glm.hpp>
#include "TestModel.h"

#define REF_API extern "C" __attribute__((visibility("default")))

static inline glm::vec3 L3(const float *p) { return glm::vec3(p[0], p[1], p[2]); }
static inline void S3(float *p, const glm::vec3 &v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
static inline glm::mat3 LM(const float *m)
{
    glm::mat3 r;
    for (int c = 0; c < 3; c++) for (int k = 0; k < 3; k++) r[c][k] = m[c * 3 + k];
    return r;
}

/* LoadTestModel as 30 x 15 floats {v0 v1 v2 normal color}; also reports sizeof(Triangle), sizeof(Light). */
REF_API int ref_load_test_model(float *out15, int *sizeof_triangle, int *sizeof_light)
{
    std::vector<Triangle> t;
    LoadTestModel(t);
    for (size_t i = 0; i < t.size(); i++) {
        S3(out15 + i * 15 + 0, t[i].v0); S3(out15 + i * 15 + 3, t[i].v1); S3(out15 + i * 15 + 6, t[i].v2);
        S3(out15 + i * 15 + 9, t[i].normal); S3(out15 + i * 15 + 12, t[i].color);
    }
    if (sizeof_triangle) *sizeof_triangle = (int)sizeof(Triangle);
    if (sizeof_light) *sizeof_light = (int)sizeof(Light);
    return (int)t.size();
}

/* Triangle::ComputeNormal on arbitrary vertices */
REF_API void ref_triangle_normal(const float *v0, const float *v1, const float *v2, float *n)
{
    Triangle t(L3(v0), L3(v1), L3(v2), glm::vec3(0, 0, 0));
    S3(n, t.normal);
}

REF_API float ref_glm_dot(const float *a, const float *b) { return glm::dot(L3(a), L3(b)); }
REF_API void ref_glm_cross(const float *a, const float *b, float *o) { S3(o, glm::cross(L3(a), L3(b))); }
REF_API void ref_glm_normalize(const float *a, float *o) { S3(o, glm::normalize(L3(a))); }
REF_API float ref_glm_distance(const float *a, const float *b) { return glm::distance(L3(a), L3(b)); }
REF_API void ref_glm_mat3_mul_vec(const float *m, const float *v, float *o) { S3(o, LM(m) * L3(v)); }
REF_API void ref_glm_vec_mul_mat3(const float *v, const float *m, float *o) { S3(o, L3(v) * LM(m)); }
REF_API void ref_glm_mat3_inverse(const float *m, float *o)
{
    glm::mat3 r = glm::inverse(LM(m));
    for (int c = 0; c < 3; c++) for (int k = 0; k < 3; k++) o[c * 3 + k] = r[c][k];
}
REF_API void ref_glm_vec_div_scalar(const float *a, float s, float *o) { S3(o, L3(a) / s); }
REF_API float ref_glm_clamp(float x, float lo, float hi) { return glm::clamp(x, lo, hi); }
/* vec4 * mat4 as used by the cull step (rasteriser.cpp:431-433) */
REF_API void ref_glm_vec4_mul_mat4(const float *v, const float *m, float *o)
{
    glm::mat4 M;
    for (int c = 0; c < 4; c++) for (int k = 0; k < 4; k++) M[c][k] = m[c * 4 + k];
    glm::vec4 r = glm::vec4(v[0], v[1], v[2], v[3]) * M;
    o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
}
