// mirt_draw.hpp -- host-side mirror of the reference's Draw() surface on top of the C-ABI (include/mirt.h).
//
// The reference renderers keep their state in globals and expose one entry point per frame, `void Draw()`
// (raytracer/Source/raytracer.cpp:104,547; rasteriser/Source/rasteriser.cpp:86,461), which ends by pushing
// pixels through PutPixelSDL into the SDL surface (raytracer.cpp:608-656, SDLauxiliary.h:70-81).  This header
// keeps that shape: the same state, under the same names, and a Draw() that (1) marshals the state into the
// POD structs of mirt.h, (2) calls the HIP path, (3) lets it store the XRGB words straight into the surface's
// `pixels` honouring `pitch`.  No SDL types are needed: a Surface is {pixels, w, h, pitch} exactly as the
// fields of SDL_Surface the reference touches (SDLauxiliary.h:72-79).
//
// The vec3 / mat3 here are layout-compatible with glm::vec3 / glm::mat3 (3 and 9 packed floats, column-major),
// so inside the reference the globals can be passed as they are.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mirt.h"

namespace mirt_host {

struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
struct mat3 {            // column-major: m[c][r], like glm::mat3
    float m[3][3];
    explicit mat3(float d = 0.0f) { std::memset(m, 0, sizeof m); m[0][0] = m[1][1] = m[2][2] = d; }
    float *operator[](int c) { return m[c]; }
    const float *operator[](int c) const { return m[c]; }
};

// class Triangle of the ray tracer (raytracer/Source/TestModel.h:11-32): 15 packed floats, sizeof == 60
struct Triangle { vec3 v0, v1, v2, normal, color; };
static_assert(sizeof(Triangle) == 60, "Triangle must be 15 packed floats");
// class Triangle of the rasteriser adds `bool isCulled` (rasteriser/Source/TestModel.h:11-33): sizeof == 64
struct RasterTriangle { vec3 v0, v1, v2, normal, color; bool isCulled; };
static_assert(sizeof(RasterTriangle) == 64, "rasteriser Triangle is 64 bytes");
// class Light (TestModel.h:35-45)
struct Light { vec3 position, color; float intensity; };
static_assert(sizeof(Light) == sizeof(mirt_light), "Light must match mirt_light");

// the fields of SDL_Surface that PutPixelSDL uses (SDLauxiliary.h:72-79)
struct Surface { uint32_t *pixels; int w, h; int pitch; };

inline void check(int rc, const char *what)
{
    if (rc != MIRT_OK) throw std::runtime_error(std::string(what) + ": " + mirt_last_error());
}

// cameraRot as Update() rebuilds it every loop (raytracer.cpp:377-382, rasteriser.cpp:378-383)
inline void update_camera_rot(mat3 &cameraRot, float yaw)
{
    const float c = std::cos(yaw), s = std::sin(yaw);      // float overloads, as in the reference
    cameraRot[0][0] = c; cameraRot[0][2] = s; cameraRot[2][0] = -s; cameraRot[2][2] = c;
}

inline mirt_view make_view(const vec3 &cameraPos, const mat3 &cameraRot, float focalLength, int W, int H)
{
    mirt_view v;
    v.pos[0] = cameraPos.x; v.pos[1] = cameraPos.y; v.pos[2] = cameraPos.z;
    std::memcpy(v.rot, cameraRot.m, sizeof v.rot);
    v.focal = focalLength;
    v.width = W; v.height = H;
    return v;
}

// ---- the ray tracer's globals and Draw() (raytracer.cpp:28-98, 547-606) ---------------------------------
struct RayTracer {
    std::vector<Triangle> triangles;                 // :28
    int NUM_LIGHTS = 0;                              // :47
    Light lights[MIRT_MAX_LIGHTS];                   // :48
    int SCREEN_WIDTH = 500, SCREEN_HEIGHT = 500;     // :67-68
    float focalLength = 250.0f;                      // :69
    vec3 cameraPos = vec3(0.0f, 0.0f, -2.0f);        // :70
    mat3 cameraRot = mat3(0.0f);                     // :73
    float yaw = 0.0f;                                // :74
    vec3 indirectLight = vec3(0.2f, 0.2f, 0.2f);     // :81  0.2f*vec3(1,1,1)
    Surface screen = { nullptr, 0, 0, 0 };           // :76
    bool isUpdated = true;                           // :78
    bool scene_dirty = true;                         // set when `triangles` changes
    bool AA_ENABLED = false;                         // :37
    int AA_SAMPLES = 3;                              // :38
    bool SOFT_SHADOWS_ENABLED = false;               // :40
    int SOFT_SHADOWS_SAMPLES = 16;                   // :41
    bool DOF_ENABLED = false;                        // :43
    int DOF_KERNEL_SIZE = 8;                         // :44
    float FOCAL_LENGTH = 1.3f;                       // :45
    vec3 randomPositions[256];                       // :84

    static float RandomNumber() { return (float)(((double)std::rand() / (RAND_MAX)) - 0.5f); }   // :260-263

    void AddLight(vec3 position, vec3 color, float intensity)     // :180-193
    {
        lights[NUM_LIGHTS].position = position; lights[NUM_LIGHTS].color = color; lights[NUM_LIGHTS].intensity = intensity;
        for (int i = 0; i < SOFT_SHADOWS_SAMPLES; i++) {          // jittered copies for soft shadows (:186-190)
            // the reference passes three RandomNumber() calls as arguments of one constructor call; g++ evaluates
            // them right to left, which is spelled out here so the result does not depend on this file's compiler
            const float rz = RandomNumber(), ry = RandomNumber(), rx = RandomNumber();
            randomPositions[(NUM_LIGHTS * SOFT_SHADOWS_SAMPLES) + i] = vec3(position.x + (rx * 0.08f), position.y + (ry * 0.08f), position.z + (rz * 0.08f));
        }
        NUM_LIGHTS++;
    }
    void LoadTestModel()                              // TestModel.h:51-192
    {
        triangles.resize(30);
        check(mirt_scene_cornell(&triangles[0].v0.x) == 30 ? MIRT_OK : MIRT_ERR_INVALID_ARGUMENT, "mirt_scene_cornell");
        scene_dirty = true;
    }
    void Update() { update_camera_rot(cameraRot, yaw); }          // the camera part of :329-545

    // Draw(): what raytracer.cpp:547-656 does, on the GPU.  Interior pixels of screen.pixels are written,
    // the 1-pixel border is left as it was (raytracer.cpp:618-620).
    void Draw()
    {
        const mirt_view view = marshal();
        check(mirt_raytrace(&view, reinterpret_cast<const mirt_light *>(lights), NUM_LIGHTS, &indirectLight.x,
                            MIRT_RT_AUTO, screen.pixels, screen.pitch, nullptr, nullptr), "mirt_raytrace");
    }
    // Draw() for a loop that presents one surface while the next one is drawn: the frame is queued into `target` (inside
    // a surface registered with mirt_surface_register) and the call returns; Present() is the loop's SDL_UpdateRect (:653).
    // With mirt_set_frames_in_flight(2) and two surfaces used in turn, frame i travels to the host while frame i + 1 renders.
    void DrawAsync(const Surface &target)
    {
        const mirt_view view = marshal();
        check(mirt_raytrace_async(&view, reinterpret_cast<const mirt_light *>(lights), NUM_LIGHTS, &indirectLight.x,
                                  MIRT_RT_AUTO, target.pixels, target.pitch), "mirt_raytrace_async");
    }
    static void Present() { check(mirt_sync(), "mirt_sync"); }

private:
    mirt_view marshal()                                            // the globals -> the POD arguments of mirt.h
    {
        if (scene_dirty) {
            check(mirt_scene_upload(&triangles[0].v0.x, nullptr, (int)triangles.size()), "mirt_scene_upload");
            scene_dirty = false;
        }
        check(mirt_set_antialiasing(AA_ENABLED ? AA_SAMPLES : 1), "mirt_set_antialiasing");   // realSamples (:549-554)
        if (SOFT_SHADOWS_ENABLED)                                  // DirectLight's `samples` (:272-275)
            check(mirt_set_soft_shadows(SOFT_SHADOWS_SAMPLES, &randomPositions[0].x, NUM_LIGHTS * SOFT_SHADOWS_SAMPLES), "mirt_set_soft_shadows");
        else
            check(mirt_set_soft_shadows(1, nullptr, 0), "mirt_set_soft_shadows");
        check(mirt_set_depth_of_field(DOF_ENABLED ? DOF_KERNEL_SIZE : 0, FOCAL_LENGTH), "mirt_set_depth_of_field");   // CalculateDOF (:608-646)
        return make_view(cameraPos, cameraRot, focalLength, SCREEN_WIDTH, SCREEN_HEIGHT);
    }
};

// ---- the rasteriser's globals, cull step and Draw() (rasteriser.cpp:22-80, 404-447, 461-529) -------------
struct Rasteriser {
    std::vector<RasterTriangle> triangles;            // :64
    int NUM_LIGHTS = 0;
    Light lights[MIRT_MAX_LIGHTS];
    bool BACKFACE_CULLING_ENABLED = true, FRUSTUM_CULLING_ENABLED = true;   // :25-26
    int SCREEN_WIDTH = 500, SCREEN_HEIGHT = 500;      // :35-36
    vec3 cameraPos = vec3(0.0f, 0.0f, -3.0f);         // :39
    mat3 cameraRot = mat3(0.0f);                      // :40
    float focalLength = 500.0f;                       // :41
    float yaw = 0.0f;
    vec3 indirectLightPowerPerArea = vec3(0.2f, 0.2f, 0.2f);   // :47
    bool DOF_ENABLED = false;                         // :29
    int DOF_KERNEL_SIZE = 8;                          // :30
    float FOCAL_LENGTH = 1.9f;                        // :31
    Surface screen = { nullptr, 0, 0, 0 };
    bool isUpdated = true;
    bool scene_dirty = true;
    bool culled_dirty = true;                        // host-side cull flags changed since the last upload (Update())
    std::vector<float> packed;                        // 15-float view of `triangles` for the ABI
    std::vector<uint8_t> culled;

    void AddLight(vec3 position, vec3 color, float intensity)
    {
        lights[NUM_LIGHTS].position = position; lights[NUM_LIGHTS].color = color; lights[NUM_LIGHTS].intensity = intensity;
        NUM_LIGHTS++;
    }
    void LoadTestModel()
    {
        float t[30 * 15];
        check(mirt_scene_cornell(t) == 30 ? MIRT_OK : MIRT_ERR_INVALID_ARGUMENT, "mirt_scene_cornell");
        triangles.resize(30);
        for (int i = 0; i < 30; i++) { std::memcpy(&triangles[i].v0.x, t + 15 * i, 60); triangles[i].isCulled = false; }
        scene_dirty = true;
    }
    // LoadSTL::LoadSTLFile (LoadSTL.cpp:17-81), the CUSTOM_MODEL branch of main() (rasteriser.cpp:106-110)
    void LoadSTLFile(const char *path = "Source/enemy1.stl")
    {
        const float colour[3] = { 0.5f, 0.5f, 0.5f };
        const int n = mirt_scene_load_stl(path, 0.05f, colour, nullptr, 0);
        check(n < 0 ? n : MIRT_OK, "mirt_scene_load_stl");
        std::vector<float> t((size_t)n * 15);
        check(std::min(0, mirt_scene_load_stl(path, 0.05f, colour, t.data(), n)), "mirt_scene_load_stl");
        triangles.resize((size_t)n);
        for (int i = 0; i < n; i++) { std::memcpy(&triangles[i].v0.x, &t[(size_t)15 * i], 60); triangles[i].isCulled = false; }
        scene_dirty = true;
    }
    // Update(): camera matrix + the cull pass (rasteriser.cpp:375-447); the clear (:183-192) happens inside Draw().
    // GPU_CULL: the pass runs on the device over the uploaded scene (mirt_cull_device) and only the flags come back.
    bool GPU_CULL = false;
    void Update()
    {
        update_camera_rot(cameraRot, yaw);
        packed.resize(triangles.size() * 15);
        for (size_t i = 0; i < triangles.size(); i++) std::memcpy(&packed[15 * i], &triangles[i].v0.x, 60);
        culled.resize(triangles.size());
        const mirt_view view = make_view(cameraPos, cameraRot, focalLength, SCREEN_WIDTH, SCREEN_HEIGHT);
        const int flags = (BACKFACE_CULLING_ENABLED ? 1 : 0) | (FRUSTUM_CULLING_ENABLED ? 2 : 0);
        if (GPU_CULL) {
            if (scene_dirty) {
                check(mirt_scene_upload(packed.data(), nullptr, (int)triangles.size()), "mirt_scene_upload");
                scene_dirty = false;
            }
            check(mirt_cull_device(&view, flags), "mirt_cull_device");
            check(mirt_scene_get_culled(culled.data(), (int)culled.size()), "mirt_scene_get_culled");
        } else {
            check(mirt_cull(packed.data(), (int)triangles.size(), &view, flags, culled.data()), "mirt_cull");
        }
        for (size_t i = 0; i < triangles.size(); i++) triangles[i].isCulled = culled[i] != 0;   // :406,412,445
        culled_dirty = !GPU_CULL;                     // host-side flags: the device copy is stale until the next Draw() uploads them
    }
    void Draw()
    {
        const mirt_view view = marshal();
        check(mirt_rasterise(&view, reinterpret_cast<const mirt_light *>(lights), NUM_LIGHTS, &indirectLightPowerPerArea.x,
                             screen.pixels, screen.pitch, nullptr, nullptr, nullptr), "mirt_rasterise");
    }
    // see RayTracer::DrawAsync (Present() is the SDL_UpdateRect of rasteriser.cpp:528)
    void DrawAsync(const Surface &target)
    {
        const mirt_view view = marshal();
        check(mirt_rasterise_async(&view, reinterpret_cast<const mirt_light *>(lights), NUM_LIGHTS, &indirectLightPowerPerArea.x,
                                   target.pixels, target.pitch), "mirt_rasterise_async");
    }
    static void Present() { check(mirt_sync(), "mirt_sync"); }

private:
    mirt_view marshal()
    {
        if (scene_dirty) {
            check(mirt_scene_upload(packed.data(), culled.data(), (int)triangles.size()), "mirt_scene_upload");
            scene_dirty = false;
            culled_dirty = false;
        } else if (culled_dirty) {
            // Host-side cull flags travel only when Update() changed them: the upload waits for every frame in flight
            // (mirt_scene_set_culled), so a loop that draws several frames per Update() keeps its overlap.  With GPU_CULL the
            // flags are already on the device and nothing is uploaded at all -- the mode DrawAsync is meant for.
            check(mirt_scene_set_culled(culled.data(), (int)culled.size()), "mirt_scene_set_culled");
            culled_dirty = false;
        }
        check(mirt_set_depth_of_field(DOF_ENABLED ? DOF_KERNEL_SIZE : 0, FOCAL_LENGTH), "mirt_set_depth_of_field");   // CalculateDOF (:484-529)
        return make_view(cameraPos, cameraRot, focalLength, SCREEN_WIDTH, SCREEN_HEIGHT);
    }
};

// SDL_SaveBMP stand-in for the demos (raytracer.cpp:175): 24-bit bottom-up BMP from XRGB8888 words
inline bool save_bmp(const Surface &s, const char *path)
{
    FILE *f = std::fopen(path, "wb");
    if (!f) return false;
    const int row = (s.w * 3 + 3) & ~3;
    const uint32_t size = 54 + (uint32_t)row * s.h;
    unsigned char hdr[54] = { 'B', 'M' };
    auto put32 = [&](int off, uint32_t v) { hdr[off] = v & 255; hdr[off + 1] = (v >> 8) & 255; hdr[off + 2] = (v >> 16) & 255; hdr[off + 3] = v >> 24; };
    put32(2, size); put32(10, 54); put32(14, 40); put32(18, (uint32_t)s.w); put32(22, (uint32_t)s.h);
    hdr[26] = 1; hdr[28] = 24; put32(34, (uint32_t)row * s.h);
    std::fwrite(hdr, 1, 54, f);
    std::vector<unsigned char> line(row, 0);
    for (int y = s.h - 1; y >= 0; y--) {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(s.pixels) + (size_t)y * s.pitch);
        for (int x = 0; x < s.w; x++) { line[3 * x] = p[x] & 255; line[3 * x + 1] = (p[x] >> 8) & 255; line[3 * x + 2] = (p[x] >> 16) & 255; }
        std::fwrite(line.data(), 1, row, f);
    }
    std::fclose(f);
    return true;
}

}  // namespace mirt_host
