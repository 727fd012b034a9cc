/*
 * oracle/ref_raster.cpp -- TEST INFRASTRUCTURE ONLY (never linked into libmirt.so).
 *
 * The reference rasteriser's hot path compiled from the reference's OWN text (see oracle/ref_rt.cpp and
 * oracle/extract_ref.py for the method).  Unmodified reference text, #included from git-ignored oracle/_ref/:
 *     globals                                  rasteriser.cpp:9-15, 22-32, 34-80
 *     Update(): camera + cull step             :377-447         InCuboid            :451-458
 *     Draw(): the triangle loop                :466-479         CalculateDOF's blur loops :486-518
 *     VertexShader :532-546   PixelShader :549-589   DrawLineSDL's body :593-612   Interpolate :615-637
 *     Bresenham :639-672   ComputePolygonRows :674-735   DrawRows :738-753   DrawPolygon :755-768
 * plus the reference's rasteriser/Source/TestModel.h (Pixel, fPixel, Vertex, their operators) and LoadSTL.cpp, where
 * they lie.  What this file adds:
 *   - the signature line of DrawLineSDL with its `SDL_Surface*` parameter typed `void*` (the body never touches it) and
 *     a null `screen` to pass to it -- no SDL type, header or function is declared;
 *   - the clear of Update() (:183-192, depthBuffer and pixelColours to 0; the reference also calls PutPixelSDL there);
 *   - a deterministic fill for `operator new`: DrawLineSDL reads Pixels of `vector<Pixel> line(pixels)` that Bresenham
 *     never wrote when x is off screen (:599, :663-669, Pixel(){} leaves them uninitialised -- SURVEY Appendix E-2).
 *     Every allocation of this library is filled with 0x5A, the byte glibc writes under MALLOC_PERTURB_=165, so those
 *     reads see x = 0x5A5A5A5A and fail the bounds test at :606.  (MALLOC_PERTURB_ itself is not enough on this glibc:
 *     the tcache path skips the fill unless GLIBC_TUNABLES=glibc.malloc.tcache_count=0 is set as well.)
 *   - C entry points that set the globals and copy results out.
 */
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>
#include <new>
#include <vector>
#include <unistd.h>
#include <glm/glm.hpp>
#include "TestModel.h"          /* the reference's rasteriser/Source/TestModel.h */
#include <omp.h>
#include "LoadSTL.cpp"          /* the reference's rasteriser/Source/LoadSTL.cpp (SDL-free) */

/* replaceable allocation functions of THIS shared object (hidden visibility: they do not leak into the process) */
__attribute__((visibility("hidden"))) void *operator new(std::size_t n)
{
    void *p = std::malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    std::memset(p, 0x5A, n);
    return p;
}
__attribute__((visibility("hidden"))) void operator delete(void *p) noexcept { std::free(p); }
__attribute__((visibility("hidden"))) void operator delete(void *p, std::size_t) noexcept { std::free(p); }

#include "ra_globals.inc"
static void *const screen = nullptr;      /* the reference: SDL_Surface* screen (:33); only ever passed through */

#include "ra_incuboid.inc"
#include "ra_vertex_shader.inc"
#include "ra_pixel_shader.inc"
#include "ra_interpolate.inc"
#include "ra_bresenham.inc"
void DrawLineSDL( void* surface, Pixel a, Pixel b, vec3 color, vec3 normal)    /* :592 with `SDL_Surface*` typed `void*` */
#include "ra_drawline_body.inc"
#include "ra_polygon_rows.inc"
#include "ra_draw_rows.inc"
#include "ra_draw_polygon.inc"

static void ref_update_camera_and_cull()
{
#include "ra_cull.inc"
}

static void ref_draw_triangles()
{
#include "ra_draw_loop.inc"
}

static void ref_blur()
{
#include "ra_dof_loop.inc"
            blurredPixels[y*SCREEN_HEIGHT+x] = finalColour;      /* the reference: PutPixelSDL( screen, x, y, finalColour ) */
        }
    }
}

#define REF_API extern "C" __attribute__((visibility("default")))
static inline vec3 L3(const float *p) { return vec3(p[0], p[1], p[2]); }
static inline void S3(float *p, const vec3 &v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

REF_API void ref_ra_size(int *w, int *h) { *w = SCREEN_WIDTH; *h = SCREEN_HEIGHT; }

REF_API void ref_ra_set_scene(const float *t15, int n)
{
    triangles.clear();
    for (int i = 0; i < n; i++) {
        const float *t = t15 + (size_t)15 * i;
        Triangle tri(L3(t), L3(t + 3), L3(t + 6), L3(t + 12));
        tri.normal = L3(t + 9);
        triangles.push_back(tri);
    }
}

REF_API int ref_ra_load_test_model(float *out15)
{
    triangles.clear();
    LoadTestModel(triangles);
    for (size_t i = 0; i < triangles.size(); i++) {
        S3(out15 + i * 15, triangles[i].v0); S3(out15 + i * 15 + 3, triangles[i].v1); S3(out15 + i * 15 + 6, triangles[i].v2);
        S3(out15 + i * 15 + 9, triangles[i].normal); S3(out15 + i * 15 + 12, triangles[i].color);
    }
    return (int)triangles.size();
}

/* LoadSTL::LoadSTLFile reads "Source/enemy1.stl" relative to the working directory (LoadSTL.cpp:27): run it from `dir` */
REF_API int ref_ra_load_stl(const char *dir, float *out15, int max_tris)
{
    char cwd[4096];
    if (!getcwd(cwd, sizeof cwd) || chdir(dir) != 0) return -1;
    std::vector<Triangle> t;
    LoadSTL loader;
    loader.LoadSTLFile(t);
    if (chdir(cwd) != 0) return -1;
    for (size_t i = 0; i < t.size() && (int)i < max_tris; i++) {
        S3(out15 + i * 15, t[i].v0); S3(out15 + i * 15 + 3, t[i].v1); S3(out15 + i * 15 + 6, t[i].v2);
        S3(out15 + i * 15 + 9, t[i].normal); S3(out15 + i * 15 + 12, t[i].color);
    }
    return (int)t.size();
}

REF_API void ref_ra_set_lights(const float *l7, int n)
{
    NUM_LIGHTS = n;
    for (int i = 0; i < n; i++) { lights[i].position = L3(l7 + 7 * i); lights[i].color = L3(l7 + 7 * i + 3); lights[i].intensity = l7[7 * i + 6]; }
}

/* camera position, yaw, focal length and cameraRot[1][1] (main() sets 1.01f, :115), cull switches and FOCAL_LENGTH; then
 * the body of Update()'s `if (isUpdated)` (:377-447): cameraRot from yaw and triangles[i].isCulled.  rot9_out / culled_out
 * (nullable) report what it computed. */
REF_API void ref_ra_update(const float *pos3, float yaw_, float focal, float rot11, int backface, int frustum, float focal_plane,
                           const float *indirect3, float *rot9_out, unsigned char *culled_out)
{
    cameraPos = L3(pos3);
    yaw = yaw_;
    focalLength = focal;
    cameraRot = mat3(0.0f);
    cameraRot[1][1] = rot11;
    BACKFACE_CULLING_ENABLED = backface != 0;
    FRUSTUM_CULLING_ENABLED = frustum != 0;
    FOCAL_LENGTH = focal_plane;
    if (indirect3) indirectLightPowerPerArea = L3(indirect3);
    ref_update_camera_and_cull();
    if (rot9_out) for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) rot9_out[c * 3 + r] = cameraRot[c][r];
    if (culled_out) for (size_t i = 0; i < triangles.size(); i++) culled_out[i] = triangles[i].isCulled ? 1 : 0;
}

/* Update()'s clear (:183-192) + Draw()'s triangle loop (:466-479), single-threaded as the reference runs it by default
 * (MULTITHREADING_ENABLED = false, :22, :131-134).  Outputs (nullable): depthBuffer[y][x], pixelColours and focalDistances
 * with the reference's own indexing y*SCREEN_HEIGHT + x. */
REF_API void ref_ra_draw(float *depth, float *rgb, float *fd)
{
    omp_set_num_threads(1);
    for (int y = 0; y < SCREEN_HEIGHT; ++y)
        for (int x = 0; x < SCREEN_WIDTH; ++x) {
            depthBuffer[y][x] = 0.0f;
            pixelColours[y * SCREEN_HEIGHT + x] = vec3(0);
            focalDistances[y * SCREEN_HEIGHT + x] = 0.0f;
        }
    ref_draw_triangles();
    const size_t n = (size_t)SCREEN_WIDTH * SCREEN_HEIGHT;
    for (int y = 0; y < SCREEN_HEIGHT; ++y)
        for (int x = 0; x < SCREEN_WIDTH; ++x) if (depth) depth[(size_t)y * SCREEN_WIDTH + x] = depthBuffer[y][x];
    for (size_t i = 0; i < n; i++) {
        if (rgb) S3(rgb + 3 * i, pixelColours[i]);
        if (fd) fd[i] = focalDistances[i];
    }
}

REF_API void ref_ra_vertex_shader(const float *v3, int *x, int *y, float *zinv, float *pos3d)
{
    Vertex v; v.position = L3(v3);
    Pixel p;
    VertexShader(v, p);
    *x = p.x; *y = p.y; *zinv = p.zinv; S3(pos3d, p.pos3d);
}

/* Interpolate(a, b, result(n)): pixels as {x, y} ints + {zinv, pos3d.xyz} floats */
REF_API void ref_ra_interpolate(const int *axy, const float *az4, const int *bxy, const float *bz4, int n, int *out_xy, float *out_z4)
{
    Pixel a(axy[0], axy[1], az4[0], L3(az4 + 1)), b(bxy[0], bxy[1], bz4[0], L3(bz4 + 1));
    std::vector<Pixel> r(n);
    Interpolate(a, b, r);
    for (int i = 0; i < n; i++) { out_xy[2 * i] = r[i].x; out_xy[2 * i + 1] = r[i].y; out_z4[4 * i] = r[i].zinv; S3(out_z4 + 4 * i + 1, r[i].pos3d); }
}

/* ComputePolygonRows of three vertex pixels; returns ROWS, writes up to max_rows left / right pixels */
REF_API int ref_ra_polygon_rows(const int *xy6, const float *z12, int max_rows, int *left_xy, float *left_z4, int *right_xy, float *right_z4)
{
    std::vector<Pixel> vp(3), l, r;
    for (int i = 0; i < 3; i++) vp[i] = Pixel(xy6[2 * i], xy6[2 * i + 1], z12[4 * i], L3(z12 + 4 * i + 1));
    ComputePolygonRows(vp, l, r);
    for (size_t i = 0; i < l.size() && (int)i < max_rows; i++) {
        left_xy[2 * i] = l[i].x; left_xy[2 * i + 1] = l[i].y; left_z4[4 * i] = l[i].zinv; S3(left_z4 + 4 * i + 1, l[i].pos3d);
        right_xy[2 * i] = r[i].x; right_xy[2 * i + 1] = r[i].y; right_z4[4 * i] = r[i].zinv; S3(right_z4 + 4 * i + 1, r[i].pos3d);
    }
    return (int)l.size();
}

/* the blur loops of CalculateDOF over given pixelColours / focalDistances (interior pixels of `out` are written) */
REF_API void ref_ra_blur(const float *rgb, const float *fd, int kernel, float *out)
{
    const size_t n = (size_t)SCREEN_WIDTH * SCREEN_HEIGHT;
    DOF_ENABLED = kernel > 1; if (kernel > 1) DOF_KERNEL_SIZE = kernel;
    omp_set_num_threads(1);
    for (size_t i = 0; i < n; i++) { pixelColours[i] = L3(rgb + 3 * i); focalDistances[i] = fd[i]; blurredPixels[i] = vec3(0.0f); }
    ref_blur();
    for (int y = 1; y < SCREEN_HEIGHT - 1; y++)
        for (int x = 1; x < SCREEN_WIDTH - 1; x++) S3(out + 3 * ((size_t)y * SCREEN_HEIGHT + x), blurredPixels[y * SCREEN_HEIGHT + x]);
}
