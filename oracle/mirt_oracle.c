/*
 * oracle/mirt_oracle.c -- CPU restatement of the reference's per-pixel render paths.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP kernels in
 * cpp-raytracer-rasterizer_amd/csrc/ and the timed "cpu_baseline" leg of bench.py.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product (libmirt.so)
 * never links, loads or falls back to it.
 *
 * PINNING.  The reference's render translation units (raytracer/Source/raytracer.cpp,
 * rasteriser/Source/rasteriser.cpp) #include <SDL.h>; SDL 1.2 is not installed in this image and no
 * stand-in header is written, so they cannot be built AS A WHOLE.  Their hot-path functions contain no SDL
 * call, though, and are compiled from the reference's own text.  The restatement is pinned by
 *   (1) the reference's own text of ClosestIntersection, DirectLight, AddLight / RandomNumber, Draw(),
 *       CalculateDOF's blur loops, the rasteriser's Update() cull step, InCuboid, VertexShader, PixelShader,
 *       Interpolate, Bresenham, ComputePolygonRows, DrawRows, DrawLineSDL's body, DrawPolygon, Draw()'s
 *       triangle loop and LoadSTL: oracle/extract_ref.py copies those line ranges verbatim into a scratch
 *       directory, oracle/ref_rt.cpp / ref_raster.cpp compile them against the reference's TestModel.h and
 *       vendored GLM into oracle/_ref/ (500x500, and the reference's own REALTIME 150x150 build), and
 *       tests/test_oracle_ref_render.py compares every function with this file BIT FOR BIT on seeded
 *       inputs (yaw != 0, several lights, soups, soft shadows, supersampling, off-screen spans);
 *   (2) recorded outputs of (1) for the frames of tests/ref_cases.py (tests/golden/ref_render.json, made by
 *       tests/golden/make_golden.py), checked by tests/test_golden_ref_render.py wherever the reference is
 *       absent -- the recipe reproduces the five float-buffer hashes of SURVEY.md Appendix C;
 *   (3) SURVEY.md Appendix C itself (tests/golden/survey_appendix_c.json, tests/test_oracle_pin.py): hashes
 *       recorded from the whole unmodified program, the only pin of PutPixelSDL's 8-bit words (its text
 *       names an SDL type, SDLauxiliary.h:70-81) and of the rasteriser's surface clear;
 *   (4) TestModel.h + the vendored GLM compiled in place (oracle/_ref/libref_model.so,
 *       tests/test_oracle_ref_model.py).
 * Not pinned by any of them: non-square frames (the reference's row stride is SCREEN_HEIGHT, E-1) and
 * frame sizes other than 500x500 / 150x150 (SCREEN_WIDTH is a compile-time constant of the reference).
 *
 * Arithmetic contract: IEEE binary32, no FMA contraction (build with
 * -ffp-contract=off, no -ffast-math), correctly rounded / and sqrtf, GLM 0.9.7.2 operation order
 * (raytracer/glm/detail/func_geometric.inl:65-72,94-115,133-159; type_mat3x3.inl:36-56,506-522;
 * type_vec3.inl:300-308,703-709; func_common.inl:409-456).
 *
 * Deliberate, documented divergences from the reference:
 *   E-1  per-pixel arrays use row stride W (the reference uses SCREEN_HEIGHT; identical when W==H);
 *   E-2  raster fragments the reference leaves uninitialised (x outside [0,W)) are skipped;
 *   E-11 triangleIndex is -1 on a miss (uninitialised in the reference).
 */
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 ld3(const float *p) { return V(p[0], p[1], p[2]); }
static inline void st3(float *p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline v3 add3(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale3(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 div3s(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }   /* type_vec3.inl:703-709 */
static inline v3 neg3(v3 a) { return V(-a.x, -a.y, -a.z); }

/* glm::dot, func_geometric.inl:65-72: products first, then (x + y) + z */
static inline float dot3(v3 a, v3 b)
{
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
/* glm::cross, func_geometric.inl:133-142 */
static inline v3 cross3(v3 x, v3 y)
{
    return V(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
/* glm::length / glm::distance, func_geometric.inl:94-115: distance(p0,p1) = length(p1 - p0) */
static inline float length3(v3 a) { return sqrtf(dot3(a, a)); }
static inline float distance3(v3 p0, v3 p1) { return length3(sub3(p1, p0)); }
/* glm::normalize, func_geometric.inl:153-159: x * inversesqrt(dot(x,x)), inversesqrt = 1/sqrt */
static inline v3 normalize3(v3 a) { return scale3(a, 1.0f / sqrtf(dot3(a, a))); }

/* mat3 is column-major as in GLM: m[c*3+r] == m[c][r] */
/* mat3 * vec3, type_mat3x3.inl:506-513 */
static inline v3 mat3_mul_vec(const float *m, v3 v)
{
    return V(m[0] * v.x + m[3] * v.y + m[6] * v.z,
             m[1] * v.x + m[4] * v.y + m[7] * v.z,
             m[2] * v.x + m[5] * v.y + m[8] * v.z);
}
/* vec3 * mat3, type_mat3x3.inl:515-522 */
static inline v3 vec_mul_mat3(v3 v, const float *m)
{
    return V(m[0] * v.x + m[1] * v.y + m[2] * v.z,
             m[3] * v.x + m[4] * v.y + m[5] * v.z,
             m[6] * v.x + m[7] * v.y + m[8] * v.z);
}
/* glm::inverse(mat3), type_mat3x3.inl:36-56 (cofactors times OneOverDeterminant) */
static void mat3_inverse(const float *m, float *inv)
{
#define M(c, r) m[(c) * 3 + (r)]
    float ood = 1.0f / (+M(0, 0) * (M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2))
                        - M(1, 0) * (M(0, 1) * M(2, 2) - M(2, 1) * M(0, 2))
                        + M(2, 0) * (M(0, 1) * M(1, 2) - M(1, 1) * M(0, 2)));
    inv[0 * 3 + 0] = +(M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2)) * ood;
    inv[1 * 3 + 0] = -(M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2)) * ood;
    inv[2 * 3 + 0] = +(M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1)) * ood;
    inv[0 * 3 + 1] = -(M(0, 1) * M(2, 2) - M(2, 1) * M(0, 2)) * ood;
    inv[1 * 3 + 1] = +(M(0, 0) * M(2, 2) - M(2, 0) * M(0, 2)) * ood;
    inv[2 * 3 + 1] = -(M(0, 0) * M(2, 1) - M(2, 0) * M(0, 1)) * ood;
    inv[0 * 3 + 2] = +(M(0, 1) * M(1, 2) - M(1, 1) * M(0, 2)) * ood;
    inv[1 * 3 + 2] = -(M(0, 0) * M(1, 2) - M(1, 0) * M(0, 2)) * ood;
    inv[2 * 3 + 2] = +(M(0, 0) * M(1, 1) - M(1, 0) * M(0, 1)) * ood;
#undef M
}

/* float -> int as the reference's x86-64 build does it (cvttss2si): truncation, and the
 * "integer indefinite" INT_MIN for NaN / out-of-range (formally UB in C++). */
static inline int f2i(float f)
{
    if (!(f > -2147483904.0f && f < 2147483648.0f)) return INT_MIN;
    return (int)f;
}

/* ------------------------------------------------------------------------------------------ */
/* helpers exported for tests                                                                  */

/* FNV-1a-64 over raw bytes with a caller-chosen offset basis.  SURVEY Appendix C's recorded hashes were
 * produced with basis 1469598103934665603 (the standard 14695981039346656037 short of its last digit);
 * that value was recovered by matching the recorded index-map hash and then confirmed on the others. */
ORACLE_API uint64_t mirt_oracle_fnv1a64(const void *data, size_t n, uint64_t basis)
{
    const unsigned char *p = (const unsigned char *)data;
    uint64_t h = basis;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}

/* cameraRot as the reference's Update() builds it from yaw (raytracer.cpp:377-382,
 * rasteriser.cpp:378-383): zero matrix, [1][1] preset by main() (1.0f ray tracer :162,
 * 1.01f rasteriser :115), c = cosf(yaw), s = sinf(yaw). */
ORACLE_API void mirt_oracle_rot_from_yaw(float yaw, float m11, float *rot9)
{
    float c = cosf(yaw), s = sinf(yaw);
    memset(rot9, 0, 9 * sizeof(float));
    rot9[1 * 3 + 1] = m11;
    rot9[0 * 3 + 0] = c;
    rot9[0 * 3 + 2] = s;
    rot9[2 * 3 + 0] = -s;
    rot9[2 * 3 + 2] = c;
}

ORACLE_API void mirt_oracle_mat3_inverse(const float *m, float *inv) { mat3_inverse(m, inv); }
ORACLE_API void mirt_oracle_mat3_mul_vec(const float *m, const float *v, float *o) { st3(o, mat3_mul_vec(m, ld3(v))); }
ORACLE_API void mirt_oracle_vec_mul_mat3(const float *v, const float *m, float *o) { st3(o, vec_mul_mat3(ld3(v), m)); }
ORACLE_API void mirt_oracle_normalize(const float *v, float *o) { st3(o, normalize3(ld3(v))); }
ORACLE_API void mirt_oracle_cross(const float *a, const float *b, float *o) { st3(o, cross3(ld3(a), ld3(b))); }
ORACLE_API float mirt_oracle_dot(const float *a, const float *b) { return dot3(ld3(a), ld3(b)); }
ORACLE_API float mirt_oracle_distance(const float *a, const float *b) { return distance3(ld3(a), ld3(b)); }

/* ------------------------------------------------------------------------------------------ */
/* scene model: Triangle = 15 floats {v0, v1, v2, normal, color} (raytracer/Source/TestModel.h:11-32) */

/* Triangle::ComputeNormal, TestModel.h:26-31: normalize(cross(e2, e1)) -- note operand order */
static void compute_normal(float *t)
{
    v3 v0 = ld3(t), v1 = ld3(t + 3), v2 = ld3(t + 6);
    v3 e1 = sub3(v1, v0), e2 = sub3(v2, v0);
    st3(t + 9, normalize3(cross3(e2, e1)));
}
static void put_tri(float *t, v3 a, v3 b, v3 c, v3 col)
{
    st3(t, a); st3(t + 3, b); st3(t + 6, c); st3(t + 12, col);
    compute_normal(t);
}

/* LoadTestModel, raytracer/Source/TestModel.h:51-192 (the rasteriser copy builds the same
 * geometry).  Returns 30. */
ORACLE_API int mirt_oracle_cornell(float *tris15)
{
    const v3 red = V(0.75f, 0.15f, 0.15f), yellow = V(0.75f, 0.75f, 0.15f), green = V(0.15f, 0.75f, 0.15f),
             cyan = V(0.15f, 0.75f, 0.75f), blue = V(0.15f, 0.15f, 0.75f), purple = V(0.75f, 0.15f, 0.75f),
             white = V(0.75f, 0.75f, 0.75f);
    const float L = 555;
    float *t = tris15;
    int n = 0;
#define TRI(a, b, c, col) do { put_tri(t + 15 * n, a, b, c, col); n++; } while (0)
    v3 A = V(L, 0, 0), B = V(0, 0, 0), C = V(L, 0, L), D = V(0, 0, L);
    v3 E = V(L, L, 0), F = V(0, L, 0), G = V(L, L, L), H = V(0, L, L);
    TRI(C, B, A, green);  TRI(C, D, B, green);      /* floor      :81-82 */
    TRI(A, E, C, purple); TRI(C, E, G, purple);     /* left wall  :85-86 */
    TRI(F, B, D, yellow); TRI(H, F, D, yellow);     /* right wall :89-90 */
    TRI(E, F, G, cyan);   TRI(F, H, G, cyan);       /* ceiling    :93-94 */
    TRI(G, D, C, white);  TRI(G, H, D, white);      /* back wall  :97-98 */

    A = V(290, 0, 114); B = V(130, 0, 65); C = V(240, 0, 272); D = V(82, 0, 225);         /* short block :103-111 */
    E = V(290, 165, 114); F = V(130, 165, 65); G = V(240, 165, 272); H = V(82, 165, 225);
    TRI(E, B, A, red); TRI(E, F, B, red);
    TRI(F, D, B, red); TRI(F, H, D, red);
    TRI(H, C, D, red); TRI(H, G, C, red);
    TRI(G, E, C, red); TRI(E, A, C, red);
    TRI(G, F, E, red); TRI(G, H, F, red);

    A = V(423, 0, 247); B = V(265, 0, 296); C = V(472, 0, 406); D = V(314, 0, 456);       /* tall block :136-144 */
    E = V(423, 330, 247); F = V(265, 330, 296); G = V(472, 330, 406); H = V(314, 330, 456);
    TRI(E, B, A, blue); TRI(E, F, B, blue);
    TRI(F, D, B, blue); TRI(F, H, D, blue);
    TRI(H, C, D, blue); TRI(H, G, C, blue);
    TRI(G, E, C, blue); TRI(E, A, C, blue);
    TRI(G, F, E, blue); TRI(G, H, F, blue);
#undef TRI
    /* scale to [-1,1]^3, flip x and y, recompute normals (:172-191).  `2/L` is int/float = float. */
    const float k = 2 / L;
    for (int i = 0; i < n; i++) {
        float *p = t + 15 * i;
        for (int v = 0; v < 3; v++) {
            float *q = p + 3 * v;
            q[0] *= k; q[1] *= k; q[2] *= k;
            q[0] -= 1.0f; q[1] -= 1.0f; q[2] -= 1.0f;
            q[0] *= -1; q[1] *= -1;
        }
        compute_normal(p);
    }
    return n;
}

/* Synthetic triangle soup (ours -- the reference has no generator):
 * mt19937(seed); u = (x >> 8) * 2^-24; per triangle 12 draws in this order: centre xyz, edge a xyz,
 * edge b xyz, colour rgb.  centre = 2u-1, edge = s*(2u-1), colour = 0.15+0.6u, all in float. */
typedef struct { uint32_t mt[624]; int idx; } mt19937_t;
static void mt_seed(mt19937_t *g, uint32_t seed)
{
    g->mt[0] = seed;
    for (int i = 1; i < 624; i++) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}
static uint32_t mt_next(mt19937_t *g)
{
    if (g->idx >= 624) {
        for (int i = 0; i < 624; i++) {
            uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}
static inline float mt_unit(mt19937_t *g) { return (float)(mt_next(g) >> 8) * (1.0f / 16777216.0f); }

ORACLE_API void mirt_oracle_soup(uint32_t seed, int n, float s, float *tris15)
{
    mt19937_t g;
    mt_seed(&g, seed);
    for (int i = 0; i < n; i++) {
        float u[12];
        for (int k = 0; k < 12; k++) u[k] = mt_unit(&g);
        v3 c = V(2.0f * u[0] - 1.0f, 2.0f * u[1] - 1.0f, 2.0f * u[2] - 1.0f);
        v3 a = V(s * (2.0f * u[3] - 1.0f), s * (2.0f * u[4] - 1.0f), s * (2.0f * u[5] - 1.0f));
        v3 b = V(s * (2.0f * u[6] - 1.0f), s * (2.0f * u[7] - 1.0f), s * (2.0f * u[8] - 1.0f));
        v3 col = V(0.15f + 0.6f * u[9], 0.15f + 0.6f * u[10], 0.15f + 0.6f * u[11]);
        put_tri(tris15 + (size_t)15 * i, c, add3(c, a), add3(c, b), col);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* ray tracer                                                                                  */

typedef struct { v3 position; float distance; int index; } hit_t;   /* struct Intersection, raytracer.cpp:91-96 */

/* ClosestIntersection, raytracer.cpp:202-257.  Returns "any triangle accepted". */
static int closest_intersection(v3 start, v3 dir, const float *tris15, int n, hit_t *closest)
{
    int any = 0;
    for (int i = 0; i < n; i++) {
        const float *t = tris15 + (size_t)15 * i;
        v3 v0 = ld3(t), v1 = ld3(t + 3), v2 = ld3(t + 6);
        v3 e1 = sub3(v1, v0), e2 = sub3(v2, v0), b = sub3(start, v0);       /* :216-218 */
        v3 e1e2 = cross3(e1, e2), be2 = cross3(b, e2), e1b = cross3(e1, b); /* :225-227 */
        v3 negD = neg3(dir);                                                  /* :229 */
        float e1e2b = e1e2.x * b.x + e1e2.y * b.y + e1e2.z * b.z;             /* :231-234 */
        float e1e2d = e1e2.x * negD.x + e1e2.y * negD.y + e1e2.z * negD.z;
        float be2d = be2.x * negD.x + be2.y * negD.y + be2.z * negD.z;
        float e1bd = e1b.x * negD.x + e1b.y * negD.y + e1b.z * negD.z;
        float tt = e1e2b / e1e2d, u = be2d / e1e2d, v = e1bd / e1e2d;         /* :237 */
        if (u + v <= 1.0f && u >= 0.0f && v >= 0.0f && tt >= 0.0f) {          /* :239 */
            v3 pos = add3(add3(v0, scale3(e1, u)), scale3(e2, v));            /* :241 */
            float distance = distance3(start, pos);                            /* :242 */
            if (closest->distance >= distance) {                               /* :243, ties -> later index */
                closest->position = pos;
                closest->distance = distance;
                closest->index = i;
            }
            any = 1;
        }
    }
    return any;
}

/* DirectLight, raytracer.cpp:265-327.  samples == 1: SOFT_SHADOWS_ENABLED = false, position = lights[k].position;
 * samples > 1: position = randomPositions[k*samples + counter] (:284-287), passed in as `jitter`. */
static v3 direct_light(const hit_t *i, const float *tris15, int n, const float *lights7, int nlights,
                       int samples, const float *jitter)
{
    v3 result = V(0, 0, 0), result2 = V(0, 0, 0);
    const float *tri = tris15 + (size_t)15 * i->index;
    for (int k = 0; k < nlights; k++) {
      for (int counter = 0; counter < samples; counter++) {                  /* :279 */
        const float *l = lights7 + 7 * k;
        v3 position = samples != 1 ? ld3(jitter + 3 * (k * samples + counter)) : ld3(l);
        v3 lightColor = scale3(ld3(l + 3), l[6]);                              /* :282 */
        float r = distance3(i->position, position);                             /* :294 */
        float A = (float)(4 * M_PI * (double)(r * r));                          /* :295, double product narrowed */
        v3 P = div3s(lightColor, (float)samples);                               /* :296, lightColor /= (float)samples */
        v3 rDir = normalize3(sub3(position, i->position));                      /* :298 */
        v3 nDir = normalize3(ld3(tri + 9));                                     /* :300 */
        v3 B = div3s(P, A);                                                     /* :301 */
        float d = dot3(rDir, nDir);
        float m = (d < 0.0f) ? 0.0f : d;                                        /* std::max(d, 0.0f), :304 */
        v3 D = scale3(B, m);
        hit_t j;
        j.distance = FLT_MAX; j.index = -1; j.position = V(0, 0, 0);
        if (closest_intersection(position, neg3(rDir), tris15, n, &j))          /* :310 */
            if (j.distance < r * 0.99f) D = V(0, 0, 0);                         /* :313-314 */
        result = add3(result, D);                                               /* :319 */
      }
        result2 = add3(result2, result);                                        /* :322, reference quirk E-3 */
    }
    return mul3(result2, ld3(tri + 12));                                        /* :325-326 */
}

/* One ClosestIntersection call / one DirectLight call, for the function-level pins against the reference's own text
 * (tests/test_oracle_ref_render.py).  The Intersection record (position, distance, index) is in/out. */
ORACLE_API int mirt_oracle_closest_intersection(const float *start, const float *dir, const float *tris15, int n,
                                                float *pos3, float *distance, int *index)
{
    hit_t h;
    h.position = ld3(pos3); h.distance = *distance; h.index = *index;
    const int any = closest_intersection(ld3(start), ld3(dir), tris15, n, &h);
    st3(pos3, h.position); *distance = h.distance; *index = h.index;
    return any;
}

ORACLE_API void mirt_oracle_direct_light(const float *pos3, float distance, int index, const float *tris15, int n,
                                         const float *lights7, int nlights, int samples, const float *jitter, float *out3)
{
    hit_t h;
    h.position = ld3(pos3); h.distance = distance; h.index = index;
    st3(out3, direct_light(&h, tris15, n, lights7, nlights, samples, jitter));
}

/* PutPixelSDL's colour conversion, raytracer/Source/SDLauxiliary.h:75-80 on an XRGB8888 surface. */
static inline uint32_t chan8(float c)
{
    float v = 255 * c;
    v = (v > 0.0f) ? v : 0.0f;        /* glm::max, func_common.inl:430-435 */
    v = (v < 255.0f) ? v : 255.0f;    /* glm::min, func_common.inl:409-414 */
    return (uint32_t)(uint8_t)(int)v;
}
static inline uint32_t pack_xrgb(v3 c) { return (chan8(c.x) << 16) | (chan8(c.y) << 8) | chan8(c.z); }

/*
 * Draw() + CalculateDOF() (DOF/AA/soft shadows off), raytracer.cpp:547-656, for rows [y0, y1).
 * Planes are full-frame W*H row-major (stride W, see E-1); only rows in the band are written.
 * out_xrgb rows are pitch_words apart; only interior pixels x in [1,W-2], y in [1,H-2] are
 * written (:618-620), the border keeps its previous value.  Any output may be NULL.
 * Returns the number of shadow rays traced (nlights per pixel whose primary ray hit).
 */
ORACLE_API uint64_t mirt_oracle_raytrace_ex(const float *tris15, int n, const float *cam_pos, const float *rot9,
                                            float focal, int W, int H, const float *lights7, int nlights,
                                            int samples, const float *jitter, int aa,
                                            const float *indirect, int y0, int y1, int threads,
                                            float *out_rgb, int32_t *out_index, float *out_dist, float *out_pos,
                                            uint32_t *out_xrgb, int pitch_words);

ORACLE_API uint64_t mirt_oracle_raytrace(const float *tris15, int n, const float *cam_pos, const float *rot9,
                                         float focal, int W, int H, const float *lights7, int nlights,
                                         const float *indirect, int y0, int y1, int threads,
                                         float *out_rgb, int32_t *out_index, float *out_dist, float *out_pos,
                                         uint32_t *out_xrgb, int pitch_words)
{
    return mirt_oracle_raytrace_ex(tris15, n, cam_pos, rot9, focal, W, H, lights7, nlights, 1, NULL, 1, indirect, y0, y1,
                                   threads, out_rgb, out_index, out_dist, out_pos, out_xrgb, pitch_words);
}

ORACLE_API uint64_t mirt_oracle_raytrace_soft(const float *tris15, int n, const float *cam_pos, const float *rot9,
                                              float focal, int W, int H, const float *lights7, int nlights,
                                              int samples, const float *jitter,
                                              const float *indirect, int y0, int y1, int threads,
                                              float *out_rgb, int32_t *out_index, float *out_dist, float *out_pos,
                                              uint32_t *out_xrgb, int pitch_words)
{
    return mirt_oracle_raytrace_ex(tris15, n, cam_pos, rot9, focal, W, H, lights7, nlights, samples, jitter, 1, indirect,
                                   y0, y1, threads, out_rgb, out_index, out_dist, out_pos, out_xrgb, pitch_words);
}

/* The general form.  samples > 1: soft shadows with `samples` jittered positions per light (jitter[(k*samples+i)*3],
 * what AddLight stores in randomPositions, raytracer.cpp:186-190).  aa > 1: supersampling with realSamples = aa
 * (AA_ENABLED / AA_SAMPLES, :37-38, 549-599) including the reference's quirks: closestIntersections[pixel] is NOT reset
 * between the aa*aa sub-rays, so a later sub-ray shades the nearest hit found so far (possibly an earlier sub-ray's), and
 * x1 only advances after a sub-ray that hit something.  Returns the shadow rays traced. */
ORACLE_API uint64_t mirt_oracle_raytrace_ex(const float *tris15, int n, const float *cam_pos, const float *rot9,
                                            float focal, int W, int H, const float *lights7, int nlights,
                                            int samples, const float *jitter, int aa,
                                            const float *indirect, int y0, int y1, int threads,
                                            float *out_rgb, int32_t *out_index, float *out_dist, float *out_pos,
                                            uint32_t *out_xrgb, int pitch_words)
{
    uint64_t nshadow = 0;
    const v3 camera = ld3(cam_pos), N = ld3(indirect);
    const float halfW = (float)W / 2.0f, halfH = (float)H / 2.0f;          /* (float)SCREEN_WIDTH/2.0f */
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
    /* pixels are independent (:557-602): rows and columns are shared out together, so that a band of a single row (the
     * full-size checks of BASELINE configs 3 and 5) still uses every thread */
#pragma omp parallel for collapse(2) schedule(dynamic, 16) reduction(+ : nshadow)
    for (int y = y0; y < y1; y++) {
        for (int x = 0; x < W; x++) {
            size_t px = (size_t)y * W + x;
            hit_t c;
            c.distance = FLT_MAX; c.index = -1; c.position = V(0, 0, 0);        /* Update() :335-339, once per frame */
            v3 avg = V(0, 0, 0);
            const int realSamples = aa > 1 ? aa : 1;                            /* :549-554 */
            float x1, y1;
            if (realSamples > 1) y1 = y - 0.5f; else y1 = y;                    /* :566-569 */
            for (int z = 0; z < realSamples; z++) {
                if (realSamples > 1) x1 = x - 0.5f; else x1 = x;                /* :573-576 */
                for (int z2 = 0; z2 < realSamples; z2++) {
                    v3 d = V(x1 - halfW, y1 - halfH, focal);                    /* :579 */
                    if (closest_intersection(camera, mat3_mul_vec(rot9, d), tris15, n, &c)) {   /* :580 */
                        v3 D = direct_light(&c, tris15, n, lights7, nlights, samples, jitter);   /* :583 */
                        v3 T = add3(D, N);                                      /* :584-586 */
                        v3 p = ld3(tris15 + (size_t)15 * c.index + 12);         /* :587 */
                        v3 R = mul3(p, T);                                      /* :588 */
                        avg = add3(avg, R);                                     /* :591 */
                        x1 += (1.0f / (float)(realSamples - 1));                /* :593, only after a hit */
                        nshadow += (uint64_t)nlights * (uint64_t)samples;
                    }
                }
                y1 += (1.0f / (float)(realSamples - 1));                        /* :596 */
            }
            avg = div3s(avg, (float)(realSamples * realSamples));               /* :599 */
            if (out_rgb) st3(out_rgb + 3 * px, avg);                            /* :600 */
            if (out_index) out_index[px] = c.index;
            if (out_dist) out_dist[px] = c.distance;
            if (out_pos) st3(out_pos + 3 * px, c.position);
            if (out_xrgb && x >= 1 && x < W - 1 && y >= 1 && y < H - 1)         /* :618-620, :646 */
                out_xrgb[(size_t)y * pitch_words + x] = pack_xrgb(avg);
        }
    }
    return nshadow;
}

/* AddLight's jitter (raytracer.cpp:186-190) with RandomNumber() (:260-263): for each of `samples` positions three
 * calls of ((double)rand()/RAND_MAX) - 0.5f, narrowed to float, times 0.08f, added to the light position.  Uses the
 * C library's rand() stream exactly as the reference does (call srand(1) first for the reference's default state). */
ORACLE_API void mirt_oracle_jitter(const float *light_pos, int samples, float *out)
{
    /* The three RandomNumber() calls are arguments of one constructor call, `vec3 randomPos(x + R*0.08f, y + R*0.08f,
     * z + R*0.08f)` (:188); C++ leaves their order unspecified and g++ -- the compiler of the reference's Makefile --
     * evaluates call arguments right to left, so z takes the first draw, then y, then x. */
    for (int i = 0; i < samples; i++)
        for (int c = 2; c >= 0; c--) {
            float r = (float)(((double)rand() / (RAND_MAX)) - 0.5f);
            out[3 * i + c] = light_pos[c] + (r * 0.08f);
        }
}

/* ------------------------------------------------------------------------------------------ */
/* rasteriser                                                                                  */

/*
 * The cull step of the rasteriser's Update(), rasteriser.cpp:385-447 (+ InCuboid :451-458).
 * flags: bit0 back-face culling, bit1 "frustum" culling (both on by default, :25-26).
 */
ORACLE_API void mirt_oracle_cull(const float *tris15, int n, const float *cam_pos, const float *rot9,
                                 float focal, int W, int H, int flags, uint8_t *culled)
{
    const v3 cam = ld3(cam_pos);
    v3 fVec = normalize3(vec_mul_mat3(V(0, 0, 1.0f), rot9));                     /* :385 */
    float nearz = cam.z + fVec.z * 0.1f, farz = cam.z + fVec.z * 15.0f;          /* :386 */
    float w = (float)W, h = (float)H;
    v3 t = V(0.0f, -h / 2.0f, focal), b = V(0.0f, h / 2.0f, focal);              /* :392-393 */
    float cy = dot3(t, b) / (length3(t) * length3(b));                           /* :394 */
    float rfovy = acosf(cy);                                                     /* :395 */
    float aspect = w / h;                                                        /* :397 */
    float m00 = (1.0f / tanf(rfovy / 2.0f)) / aspect;                            /* :398 */
    float m11 = (1.0f / tanf(rfovy / 2.0f));                                     /* :399 */
    float m22 = farz / (farz - nearz);                                           /* :400 */
    /* transform[3][2] is assigned twice (:401-402); the surviving value is 1.0f */
    float tr[16];
    memset(tr, 0, sizeof tr);
    tr[0 * 4 + 0] = m00; tr[1 * 4 + 1] = m11; tr[2 * 4 + 2] = m22; tr[3 * 4 + 2] = 1.0f;
    for (int i = 0; i < n; i++) {
        const float *p = tris15 + (size_t)15 * i;
        int c = 0;
        if (flags & 1)
            if (dot3(sub3(ld3(p), cam), ld3(p + 9)) > 0.0f) c = 1;               /* :408-414 */
        if ((flags & 2) && !c) {
            int inside[3];
            for (int k = 0; k < 3; k++) {
                v3 q = vec_mul_mat3(sub3(ld3(p + 3 * k), cam), rot9);            /* :423-425 */
                float v[4] = { q.x, q.y, q.z, 1.0f }, o[4];
                for (int j = 0; j < 4; j++)                                      /* vec4 * mat4, type_mat4x4.inl:664-675 */
                    o[j] = tr[j * 4 + 0] * v[0] + tr[j * 4 + 1] * v[1] + tr[j * 4 + 2] * v[2] + tr[j * 4 + 3] * v[3];
                float wd = o[3];
                float X = o[0] / wd, Y = o[1] / wd, Z = o[2] / wd;               /* :435-437 */
                inside[k] = (X >= -1.0f && X <= 1.0f && Y >= -1.0f && Y <= 1.0f && Z >= 0.0f && Z <= 1.0f);
            }
            if (!inside[0] && !inside[1] && !inside[2]) c = 1;                   /* :444-445 */
        }
        culled[i] = (uint8_t)c;
    }
}

typedef struct { int x, y; float zinv; v3 pos3d; } pixel_t;    /* struct Pixel, rasteriser/Source/TestModel.h:34-53 */

/* VertexShader, rasteriser.cpp:532-546 */
static void vertex_shader(v3 v, v3 cam, const float *rot9, float focal, int W, int H, pixel_t *p)
{
    v3 pos = vec_mul_mat3(sub3(v, cam), rot9);
    p->pos3d = div3s(pos, pos.z);
    p->zinv = 1.0f / pos.z;
    p->x = f2i((float)f2i(focal * (pos.x * p->zinv)) + ((float)W / 2.0f));
    p->y = f2i((float)f2i(focal * (pos.y * p->zinv)) + ((float)H / 2.0f));
}

ORACLE_API void mirt_oracle_vertex_shader(const float *v, const float *cam_pos, const float *rot9, float focal,
                                          int W, int H, int *x, int *y, float *zinv, float *pos3d)
{
    pixel_t p;
    vertex_shader(ld3(v), ld3(cam_pos), rot9, focal, W, H, &p);
    *x = p.x; *y = p.y; *zinv = p.zinv; st3(pos3d, p.pos3d);
}

/* Screen coordinates beyond this are outside the contract: the reference itself would size
 * std::vectors from them (UB / bad_alloc).  Such triangles are skipped by oracle and product alike. */
#define MIRT_RASTER_COORD_LIMIT (1 << 20)

/* PixelShader, rasteriser.cpp:549-589 (inverse hoisted: same operands => same bits, E-6). */
static v3 pixel_shader(v3 pos3d, float zinv, const float *invrot, v3 cam, const float *lights7, int nlights,
                       v3 indirect, v3 color, v3 normal, float *cam_distance)
{
    v3 P = div3s(pos3d, zinv);                                                   /* :557 */
    P = vec_mul_mat3(P, invrot);                                                 /* :559 */
    P = add3(P, cam);                                                            /* :560 */
    *cam_distance = distance3(P, cam);                                           /* :563, feeds focalDistances :564-565 */
    v3 result = V(0, 0, 0);
    for (int i = 0; i < nlights; i++) {
        const float *l = lights7 + 7 * i;
        v3 lightPos = ld3(l);
        float r = distance3(P, lightPos);                                        /* :574 */
        float A = (float)(4 * M_PI * (double)(r * r));                           /* :575 */
        v3 lightColor = scale3(ld3(l + 3), l[6]);                                /* :576 */
        v3 rDir = normalize3(sub3(lightPos, P));                                 /* :577 */
        v3 B = div3s(lightColor, A);                                             /* :579 */
        float d = dot3(rDir, normal);
        float m = (d < 0.0f) ? 0.0f : d;                                         /* std::max, :581 */
        result = add3(result, scale3(B, m));                                     /* :581-582 */
    }
    /* currentReflectance (1,1,1) * (result + indirect) * color, :587 */
    return mul3(mul3(V(1.0f, 1.0f, 1.0f), add3(result, indirect)), color);
}

/*
 * Update()'s clear + Draw() + CalculateDOF() of the rasteriser (rasteriser.cpp:183-192, 461-529,
 * 532-768) in the single-thread triangle order (the reference default, :22).
 * out_depth (W*H, required) = depthBuffer; out_rgb (W*H*3, nullable) = pixelColours; out_index
 * (nullable) = winning triangle per pixel or -1; out_xrgb: every pixel is cleared to 0 (Update
 * paints the whole surface black, :190) then interior pixels get the resolved colour (:491-519).
 */
ORACLE_API void mirt_oracle_rasterise_ex(const float *tris15, const uint8_t *culled, int n, const float *cam_pos,
                                         const float *rot9, float focal, int W, int H, const float *lights7,
                                         int nlights, const float *indirect, float focal_plane,
                                         float *out_depth, float *out_rgb, int32_t *out_index, float *out_fd,
                                         uint32_t *out_xrgb, int pitch_words);

ORACLE_API void mirt_oracle_rasterise(const float *tris15, const uint8_t *culled, int n, const float *cam_pos,
                                      const float *rot9, float focal, int W, int H, const float *lights7,
                                      int nlights, const float *indirect,
                                      float *out_depth, float *out_rgb, int32_t *out_index,
                                      uint32_t *out_xrgb, int pitch_words)
{
    mirt_oracle_rasterise_ex(tris15, culled, n, cam_pos, rot9, focal, W, H, lights7, nlights, indirect, 0.0f,
                             out_depth, out_rgb, out_index, NULL, out_xrgb, pitch_words);
}

/* The same, also returning focalDistances = distance(pPos3d, cameraPos) - FOCAL_LENGTH of the fragment that owns each
 * pixel (rasteriser.cpp:563-565; 0 where nothing was drawn -- the reference never clears that array). */
ORACLE_API void mirt_oracle_rasterise_ex(const float *tris15, const uint8_t *culled, int n, const float *cam_pos,
                                         const float *rot9, float focal, int W, int H, const float *lights7,
                                         int nlights, const float *indirect, float focal_plane,
                                         float *out_depth, float *out_rgb, int32_t *out_index, float *out_fd,
                                         uint32_t *out_xrgb, int pitch_words)
{
    const v3 cam = ld3(cam_pos), ind = ld3(indirect);
    float invrot[9];
    mat3_inverse(rot9, invrot);
    size_t npx = (size_t)W * H;
    float *rgb = out_rgb ? out_rgb : (float *)malloc(npx * 3 * sizeof(float));
    for (size_t i = 0; i < npx; i++) out_depth[i] = 0.0f;                        /* :188 */
    memset(rgb, 0, npx * 3 * sizeof(float));                                     /* :189 */
    if (out_index) for (size_t i = 0; i < npx; i++) out_index[i] = -1;
    if (out_fd) for (size_t i = 0; i < npx; i++) out_fd[i] = 0.0f;

    pixel_t *left = NULL, *right = NULL, *edge = NULL;
    size_t cap = 0;
    for (int ti = 0; ti < n; ti++) {
        if (culled && culled[ti]) continue;                                      /* :470 */
        const float *t = tris15 + (size_t)15 * ti;
        v3 color = ld3(t + 12), normal = ld3(t + 9);
        pixel_t vp[3];
        int ok = 1;
        for (int k = 0; k < 3; k++) {
            vertex_shader(ld3(t + 3 * k), cam, rot9, focal, W, H, &vp[k]);       /* :761 */
            if (vp[k].x <= -MIRT_RASTER_COORD_LIMIT || vp[k].x >= MIRT_RASTER_COORD_LIMIT ||
                vp[k].y <= -MIRT_RASTER_COORD_LIMIT || vp[k].y >= MIRT_RASTER_COORD_LIMIT) ok = 0;
        }
        if (!ok) continue;
        /* ComputePolygonRows, :674-735 */
        int maxY = vp[0].y > vp[1].y ? vp[0].y : vp[1].y; if (vp[2].y > maxY) maxY = vp[2].y;
        int minY = vp[0].y < vp[1].y ? vp[0].y : vp[1].y; if (vp[2].y < minY) minY = vp[2].y;
        int rows = maxY - minY + 1;
        if ((size_t)rows > cap) {
            cap = (size_t)rows;
            left = (pixel_t *)realloc(left, cap * sizeof(pixel_t));
            right = (pixel_t *)realloc(right, cap * sizeof(pixel_t));
            edge = (pixel_t *)realloc(edge, cap * sizeof(pixel_t));
        }
        for (int i = 0; i < rows; i++) { left[i].x = INT_MAX; right[i].x = -INT_MAX; }
        for (int i = 0; i < 3; i++) {
            int j = (i + 1) % 3;
            pixel_t a = vp[i], b = vp[j];
            a.y -= minY; b.y -= minY;                                            /* :710-711 */
            int N = abs(vp[i].y - vp[j].y) + 1;                                  /* :713 */
            /* Interpolate, :615-637: step = (b-a)/max(N-1,1), then sequential accumulation */
            float div = (float)((N - 1) > 1 ? (N - 1) : 1);
            float sx = (float)(b.x - a.x) / div, sy = (float)(b.y - a.y) / div;
            float sz = (b.zinv - a.zinv) / div;
            v3 sp = div3s(sub3(b.pos3d, a.pos3d), div);
            float cx = (float)a.x, cyy = (float)a.y, cz = a.zinv;
            v3 cp = a.pos3d;
            for (int k = 0; k < N; k++) {
                edge[k].x = f2i(cx); edge[k].y = f2i(cyy); edge[k].zinv = cz; edge[k].pos3d = cp;
                cx += sx; cyy += sy; cz += sz; cp = add3(cp, sp);
            }
            for (int k = 0; k < N; k++) {                                        /* :716-733 */
                int row = edge[k].y;
                if (edge[k].x < left[row].x) { left[row] = edge[k]; left[row].y = edge[k].y + minY; }
                if (edge[k].x > right[row].x) { right[row] = edge[k]; right[row].y = edge[k].y + minY; }
            }
        }
        /* DrawRows, :738-753 -> DrawLineSDL, :592-612 -> Bresenham, :639-672 */
        for (int i = 0; i < rows; i++) {
            pixel_t a = left[i], b = right[i];
            if ((a.y >= H && b.y >= H) || (a.y < 0 && b.y < 0)) continue;
            int dx = b.x - a.x;
            if (dx <= 0) continue;
            float zstep = (b.zinv - a.zinv) / (float)dx;                          /* :648 */
            v3 pstep = div3s(sub3(b.pos3d, a.pos3d), (float)dx);                  /* :649 */
            int y = a.y;                                                          /* dy == 0: y never advances */
            for (int k = 0; k < dx; k++) {
                int x = a.x + 1 + k;
                if (!(x >= 0 && x < W)) continue;                                 /* :663, skipped == E-2 */
                float zinv = a.zinv + zstep * (float)k;                           /* :667 */
                v3 p3 = add3(a.pos3d, scale3(pstep, (float)k));                   /* :668 */
                if (y < H && y >= 0 && zinv > out_depth[(size_t)y * W + x]) {     /* :606 */
                    size_t px = (size_t)y * W + x;
                    out_depth[px] = zinv;                                         /* :608 */
                    float camdist;
                    st3(rgb + 3 * px, pixel_shader(p3, zinv, invrot, cam, lights7, nlights, ind, color, normal, &camdist));
                    if (out_index) out_index[px] = ti;
                    if (out_fd) out_fd[px] = camdist - focal_plane;                 /* :564-565 */
                }
            }
        }
    }
    if (out_xrgb) {
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                uint32_t w = 0;                                                   /* Update() :190 */
                if (x >= 1 && x < W - 1 && y >= 1 && y < H - 1)                   /* :491-493, :516 */
                    w = pack_xrgb(ld3(rgb + 3 * ((size_t)y * W + x)));
                out_xrgb[(size_t)y * pitch_words + x] = w;
            }
    }
    free(left); free(right); free(edge);
    if (!out_rgb) free(rgb);
}

/* ------------------------------------------------------------------------------------------ */
/* depth of field: the DOF_ENABLED branch of CalculateDOF (raytracer.cpp:613-640, rasteriser.cpp:494-513)   */

/* rgb = pixelColours, fd = focalDistances (W*H, stride W).  Resolves rows [y0,y1) into out_xrgb: interior pixels get the
 * blurred colour; border pixels are left alone, or zeroed when clear_border (the rasteriser's Update() painted the whole
 * surface black).  Tap addresses are flat indices as in the reference; a flat index outside the frame is undefined
 * behaviour there and contributes nothing here (documented divergence). */
static float *g_dof_float_out = NULL;      /* set by mirt_oracle_dof_float only (single-threaded test use) */

ORACLE_API void mirt_oracle_dof(const float *rgb, const float *fd, int W, int H, int K, int y0, int y1, int clear_border,
                                uint32_t *out_xrgb, int pitch_words)
{
    const float totalPixels = (float)(K * K);                                    /* :615 */
    const int zlo = (int)ceilf((float)K / -2.0f), zhi = (int)ceilf((float)K / 2.0f);   /* :624,626 */
    const long long npx = (long long)W * H;
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < W; x++) {
            if (!(x >= 1 && x < W - 1 && y >= 1 && y < H - 1)) {                 /* :618-620 */
                if (clear_border) out_xrgb[(size_t)y * pitch_words + x] = 0u;
                continue;
            }
            v3 fin = V(0, 0, 0);
            const float f = fd[(size_t)y * W + x];
            const float a = fabsf(f) < 1.0f ? fabsf(f) : 1.0f;                   /* std::min(abs(fd), 1.0f) */
            for (int z = zlo; z < zhi; z++)
                for (int z2 = zlo; z2 < zhi; z2++) {
                    float weighting;
                    if (z == 0 && z2 == 0) weighting = 1 - (a * ((totalPixels - 1) / totalPixels));   /* :629 */
                    else weighting = a * (1.0f / totalPixels);                   /* :631 */
                    const long long idx = (long long)(y + z) * W + (x + z2);     /* :634 */
                    v3 c = V(0, 0, 0);
                    if (idx >= 0 && idx < npx) c = ld3(rgb + 3 * idx);
                    fin = add3(fin, scale3(c, weighting));
                }
            out_xrgb[(size_t)y * pitch_words + x] = pack_xrgb(fin);              /* PutPixelSDL :646 */
            if (g_dof_float_out) st3(g_dof_float_out + 3 * ((size_t)y * W + x), fin);
        }
}

/* The same with finalColour itself (what the reference hands to PutPixelSDL) stored as floats for interior pixels. */
ORACLE_API void mirt_oracle_dof_float(const float *rgb, const float *fd, int W, int H, int K, float *out_rgb, uint32_t *scratch_xrgb)
{
    g_dof_float_out = out_rgb;
    mirt_oracle_dof(rgb, fd, W, H, K, 0, H, 0, scratch_xrgb, W);
    g_dof_float_out = NULL;
}

/* ------------------------------------------------------------------------------------------ */
/* LoadSTL::LoadSTLFile + split, rasteriser/Source/LoadSTL.cpp:17-97                            */

/* Reads an ASCII STL as the reference does: every line that contains "outer" (:35) is followed by three vertex lines
 * (:40-57), each split at single spaces with empty tokens and the token "vertex" dropped (:84-97), the first three tokens
 * through atof; then every coordinate *= -scale (:63-76; the reference's scale is 0.05f), colour (0.5, 0.5, 0.5) there
 * (:22), normal by ComputeNormal (:78).  Returns the facet count (-1: cannot open, -2: a vertex line with fewer than
 * three tokens -- an out-of-range index in the reference).  Writes at most max_tris triangles when tris15 != NULL. */
ORACLE_API int mirt_oracle_load_stl(const char *path, float scale, const float *colour3, float *tris15, int max_tris)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    int n = 0, c;
    size_t cap = 256, len;
    char *line = (char *)malloc(cap);
    for (;;) {
        /* std::getline: everything up to the next newline */
        len = 0;
        int any = 0;
        while ((c = fgetc(f)) != EOF) {
            any = 1;
            if (c == 10) break;
            if (len + 2 > cap) { cap *= 2; line = (char *)realloc(line, cap); }
            line[len++] = (char)c;
        }
        if (!any) break;
        line[len] = 0;
        if (!strstr(line, "outer")) continue;
        float v[9];
        for (int i = 0; i < 3; i++) {
            len = 0;
            while ((c = fgetc(f)) != EOF) {
                if (c == 10) break;
                if (len + 2 > cap) { cap *= 2; line = (char *)realloc(line, cap); }
                line[len++] = (char)c;
            }
            line[len] = 0;
            /* split(line, ' '): tokens between single spaces; empty ones and "vertex" are dropped */
            int nt = 0;
            char *p = line;
            for (;;) {
                char *e = strchr(p, ' ');
                size_t tl = e ? (size_t)(e - p) : strlen(p);
                if (tl > 0 && !(tl == 6 && strncmp(p, "vertex", 6) == 0)) {
                    if (nt < 3) {
                        char save = p[tl];
                        p[tl] = 0;
                        v[3 * i + nt] = (float)atof(p);
                        p[tl] = save;
                    }
                    nt++;
                }
                if (!e) break;
                p = e + 1;
            }
            if (nt < 3) { free(line); fclose(f); return -2; }
        }
        if (tris15 && n < max_tris) {
            float *t = tris15 + (size_t)15 * n;
            for (int k = 0; k < 9; k++) t[k] = v[k] * -scale;
            t[12] = colour3[0]; t[13] = colour3[1]; t[14] = colour3[2];
            compute_normal(t);
        }
        n++;
    }
    free(line);
    fclose(f);
    return n;
}
