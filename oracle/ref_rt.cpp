/*
 * oracle/ref_rt.cpp -- TEST INFRASTRUCTURE ONLY (never linked into libmirt.so).
 *
 * The reference ray tracer's hot path compiled from the reference's OWN text.  raytracer.cpp as a whole needs
 * <SDL.h> (absent in this image; no stand-in is written), but the functions on the hot path contain no SDL
 * call: oracle/extract_ref.py copies their line ranges verbatim into git-ignored oracle/_ref/*.inc and this
 * harness #includes them between the reference's own TestModel.h and the GLM it vendors (both found where
 * they lie through -I, see oracle/Makefile).  What comes from the reference, unmodified:
 *     globals + struct Intersection          raytracer.cpp:22-24, 28-75, 77-98, 103-112
 *     main()'s closestIntersections fill      :152-162        AddLight         :180-193
 *     ClosestIntersection                     :202-257        RandomNumber     :260-263
 *     DirectLight                             :265-327        Update()'s reset :335-339, camera :377-382
 *     Draw() (without its CalculateDOF call)  :547-603        CalculateDOF's blur loops :613-645
 * What this file adds: C entry points that set those globals and copy results out, the closing brace of
 * Draw(), and the store + closing braces after the blur loops (the reference calls PutPixelSDL there).
 * Built twice by oracle/Makefile: 500x500 (the default) and -DREALTIME (the reference's own 150x150 mode,
 * raytracer.cpp:59-64).  tests/test_oracle_ref_render.py compares mirt_oracle.c with it bit for bit and
 * tests/golden/make_golden.py records its outputs as fixtures for the GPU box.
 */
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>
#include <vector>
#include <glm/glm.hpp>
#include "TestModel.h"          /* the reference's raytracer/Source/TestModel.h */
#include <omp.h>

#include "rt_globals.inc"
#include "rt_random.inc"
#include "rt_addlight.inc"
#include "rt_closest.inc"
#include "rt_direct.inc"
#include "rt_draw.inc"
}   /* closes Draw(): the reference calls CalculateDOF() here (SDL lock / PutPixelSDL / SDL_UpdateRect) */

static void ref_blur()
{
#include "rt_dof_loop.inc"
            blurredPixels[y*SCREEN_HEIGHT+x] = finalColour;      /* the reference: PutPixelSDL( screen, x, y, finalColour ) */
        }
    }
}

static void ref_init_intersections()
{
    closestIntersections.clear();
#include "rt_init_intersections.inc"
}

static void ref_reset_distances()
{
#include "rt_reset.inc"
}

static void ref_camera_from_yaw()
{
#include "rt_camera.inc"
}

#define REF_API extern "C" __attribute__((visibility("default")))
static inline vec3 L3(const float *p) { return vec3(p[0], p[1], p[2]); }
static inline void S3(float *p, const vec3 &v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

REF_API void ref_rt_size(int *w, int *h, float *focal, float *cam3)
{
    *w = SCREEN_WIDTH; *h = SCREEN_HEIGHT; *focal = focalLength; S3(cam3, cameraPos);
}

/* triangles as n x 15 floats {v0 v1 v2 normal color}: the constructor recomputes the normal (TestModel.h:20-31),
 * then the given one is stored so that scenes with arbitrary normals can be pinned too */
REF_API void ref_rt_set_scene(const float *t15, int n)
{
    triangles.clear();
    for (int i = 0; i < n; i++) {
        const float *t = t15 + (size_t)15 * i;
        Triangle tri(L3(t), L3(t + 3), L3(t + 6), L3(t + 12));
        tri.normal = L3(t + 9);
        triangles.push_back(tri);
    }
    if (closestIntersections.size() != (size_t)SCREEN_WIDTH * SCREEN_HEIGHT) ref_init_intersections();
}

REF_API int ref_rt_load_test_model(float *out15)
{
    triangles.clear();
    LoadTestModel(triangles);
    for (size_t i = 0; i < triangles.size(); i++) {
        S3(out15 + i * 15, triangles[i].v0); S3(out15 + i * 15 + 3, triangles[i].v1); S3(out15 + i * 15 + 6, triangles[i].v2);
        S3(out15 + i * 15 + 9, triangles[i].normal); S3(out15 + i * 15 + 12, triangles[i].color);
    }
    if (closestIntersections.size() != (size_t)SCREEN_WIDTH * SCREEN_HEIGHT) ref_init_intersections();
    return (int)triangles.size();
}

/* camera: position, yaw (Update() builds cameraRot from it, :377-382; cameraRot[1][1] = 1 as main() sets it), focal length */
REF_API void ref_rt_set_view(const float *pos3, float yaw_, float focal, float *rot9_out)
{
    cameraPos = L3(pos3);
    yaw = yaw_;
    focalLength = focal;
    cameraRot = mat3(0.0f);
    cameraRot[1][1] = 1.0f;
    ref_camera_from_yaw();
    if (rot9_out) for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) rot9_out[c * 3 + r] = cameraRot[c][r];
}

REF_API void ref_rt_clear_lights(void) { NUM_LIGHTS = 0; }
REF_API void ref_rt_srand(unsigned seed) { srand(seed); }
/* AddLight (:180-193): also draws the SOFT_SHADOWS_SAMPLES jittered positions of the new light from rand() */
REF_API void ref_rt_add_light(const float *pos3, const float *col3, float intensity) { AddLight(L3(pos3), L3(col3), intensity); }
REF_API void ref_rt_random_positions(float *out, int count) { for (int i = 0; i < count; i++) S3(out + 3 * i, randomPositions[i]); }
REF_API void ref_rt_set_indirect(const float *c3) { indirectLight = L3(c3); }

REF_API void ref_rt_set_options(int aa_samples, int soft_samples, int dof_kernel, float focal_plane, int threads)
{
    AA_ENABLED = aa_samples > 1; if (aa_samples > 1) AA_SAMPLES = aa_samples;
    SOFT_SHADOWS_ENABLED = soft_samples > 1; if (soft_samples > 1) SOFT_SHADOWS_SAMPLES = soft_samples;
    DOF_ENABLED = dof_kernel > 1; if (dof_kernel > 1) DOF_KERNEL_SIZE = dof_kernel;
    FOCAL_LENGTH = focal_plane;
    omp_set_num_threads(threads > 0 ? threads : 1);
}

/* one ClosestIntersection call; in/out = the Intersection record (pos3, distance, index) */
REF_API int ref_rt_closest(const float *start3, const float *dir3, int is_light, float *pos3, float *distance, int *index)
{
    Intersection it;
    it.position = L3(pos3); it.distance = *distance; it.triangleIndex = *index;
    const bool any = ClosestIntersection(L3(start3), L3(dir3), triangles, it, is_light != 0, 0, 0);
    S3(pos3, it.position); *distance = it.distance; *index = it.triangleIndex;
    return any ? 1 : 0;
}

REF_API void ref_rt_direct_light(const float *pos3, float distance, int index, float *out3)
{
    Intersection it;
    it.position = L3(pos3); it.distance = distance; it.triangleIndex = index;
    S3(out3, DirectLight(it));
}

/* Update()'s reset (:335-339) + Draw() (:547-603).  Outputs (each nullable), SCREEN_WIDTH x SCREEN_HEIGHT, row stride
 * SCREEN_HEIGHT as the reference indexes them: closest-hit index (-1 where nothing was hit: the reference leaves it
 * uninitialised), distance, position, pixelColours, focalDistances. */
REF_API void ref_rt_draw(int *index, float *distance, float *position, float *rgb, float *fd)
{
    const size_t n = (size_t)SCREEN_WIDTH * SCREEN_HEIGHT;
    ref_reset_distances();
    for (size_t i = 0; i < n; i++) { closestIntersections[i].triangleIndex = -1; closestIntersections[i].position = vec3(0.0f); focalDistances[i] = 0.0f; }
    Draw();
    for (size_t i = 0; i < n; i++) {
        if (index) index[i] = closestIntersections[i].triangleIndex;
        if (distance) distance[i] = closestIntersections[i].distance;
        if (position) S3(position + 3 * i, closestIntersections[i].position);
        if (rgb) S3(rgb + 3 * i, pixelColours[i]);
        if (fd) fd[i] = focalDistances[i];
    }
}

/* the blur loops of CalculateDOF over given pixelColours / focalDistances; out = what the reference hands to PutPixelSDL
 * (interior pixels; the border of `out` is left as it is) */
REF_API void ref_rt_blur(const float *rgb, const float *fd, float *out)
{
    const size_t n = (size_t)SCREEN_WIDTH * SCREEN_HEIGHT;
    for (size_t i = 0; i < n; i++) { pixelColours[i] = L3(rgb + 3 * i); focalDistances[i] = fd[i]; blurredPixels[i] = vec3(0.0f); }
    ref_blur();
    for (int y = 1; y < SCREEN_HEIGHT - 1; y++)
        for (int x = 1; x < SCREEN_WIDTH - 1; x++) S3(out + 3 * ((size_t)y * SCREEN_HEIGHT + x), blurredPixels[y * SCREEN_HEIGHT + x]);
}
