// frame_rate.cpp -- the rate of the C-ABI as a C++ host drives it (the reference's language; bench.py drives it through ctypes, whose
// call overhead is a third of a 4.7 us frame): the reference's own scene (LoadTestModel, TestModel.h:51-192) through
// mirt_raytrace_device / mirt_cull_device + mirt_rasterise_device into device surfaces, the camera turning by one milliradian per
// frame as in bench.py (raytracer.cpp:353-383 turns it by the arrow keys), one to four frames in flight.
//   host/frame_rate [rt|raster] [W H] [frames in flight]
// Device memory comes from hipMalloc (linked against libamdhip64 for that alone); everything else is the library's C-ABI.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/mirt.h"

static void check(int rc, const char *what)
{
    if (rc != MIRT_OK) { std::fprintf(stderr, "%s: %s\n", what, mirt_last_error()); std::exit(1); }
}

static mirt_view view_of(float yaw, float m11, float z, float focal, int W, int H)
{
    mirt_view v{};
    v.pos[0] = 0.0f; v.pos[1] = 0.0f; v.pos[2] = z;
    const float c = std::cos(yaw), s = std::sin(yaw);            // cameraRot as Update() builds it (raytracer.cpp:377-382), row-major
    const float r[9] = { c, 0.0f, s, 0.0f, m11, 0.0f, -s, 0.0f, c };
    for (int i = 0; i < 9; i++) v.rot[i] = r[i];
    v.focal = focal; v.width = W; v.height = H;
    return v;
}

int main(int argc, char **argv)
{
    const std::string which = argc > 1 ? argv[1] : "rt";
    const int W = argc > 3 ? std::atoi(argv[2]) : 500, H = argc > 3 ? std::atoi(argv[3]) : 500;
    const int flight = argc > 4 ? std::atoi(argv[4]) : 4;
    const bool raster = which == "raster";
    check(mirt_init(0), "mirt_init");
    std::vector<float> tris(15 * 64);
    const int n = mirt_scene_cornell(tris.data());
    if (n <= 0) { std::fprintf(stderr, "mirt_scene_cornell: %s\n", mirt_last_error()); return 1; }
    check(mirt_scene_upload(tris.data(), nullptr, n), "mirt_scene_upload");
    check(mirt_set_frames_in_flight(flight), "mirt_set_frames_in_flight");
    const mirt_light light = { { 0.0f, -0.5f, -0.7f }, { 1.0f, 1.0f, 1.0f }, 14.0f };
    const float indirect[3] = { 0.2f, 0.2f, 0.2f };
    std::vector<mirt_view> views;
    for (int i = 0; i < 64; i++) views.push_back(view_of(0.001f * i, raster ? 1.01f : 1.0f, -3.0f, (float)H, W, H));
    void *surf[4];
    for (auto &p : surf)
        if (hipMalloc(&p, (size_t)W * H * 4) != hipSuccess) { std::fprintf(stderr, "hipMalloc failed\n"); return 1; }
    auto frame = [&](int i) {
        const mirt_view &v = views[i & 63];
        if (raster) {
            check(mirt_cull_device(&v, 3), "mirt_cull_device");
            check(mirt_rasterise_device(&v, &light, 1, indirect, 0, H, 0, surf[i & 3], W * 4, nullptr, nullptr, nullptr), "mirt_rasterise_device");
        } else {
            check(mirt_raytrace_device(&v, &light, 1, indirect, MIRT_RT_AUTO, 0, H, 0, surf[i & 3], W * 4, nullptr, nullptr), "mirt_raytrace_device");
        }
    };
    double best = 1e30;
    const int frames = 20000;
    for (int rep = 0; rep < 5; rep++) {
        for (int i = 0; i < 500; i++) frame(i);
        check(mirt_sync(), "mirt_sync");
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < frames; i++) frame(i);
        const double host = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / frames;
        check(mirt_sync(), "mirt_sync");
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / frames;
        if (dt < best) best = dt;
        if (rep == 4)
            std::printf("%s %dx%d Cornell box, %d frame(s) in flight, C++ host loop: %.2f us per frame (best of 5 x %d; the calls alone %.2f us of the "
                        "launching thread in the last run) = %.0f frames/s\n", raster ? "rasteriser" : "ray tracer", W, H, flight, best * 1e6, frames, host * 1e6, 1.0 / best);
    }
    for (auto p : surf) (void)hipFree(p);
    mirt_shutdown();
    return 0;
}
