// demo_main.cpp -- the reference's main() loops (raytracer.cpp:113-178, rasteriser.cpp:101-149) on top of
// mirt_draw.hpp, without SDL: a heap surface stands in for the window, one Update()+Draw() per "loop".
//   demo_main rt|rtsoft|rtaa|rtdof|rtasync|raster|rasterdof|rasterstl|rasterasync [width height [out.bmp [out.xrgb [model.stl]]]]
//   (rtsoft: SOFT_SHADOWS_ENABLED, rtaa: AA_ENABLED, *dof: DOF_ENABLED, rasterstl: the CUSTOM_MODEL build, culled on the GPU,
//   *async: DrawAsync() into two registered surfaces in turn with two frames in flight, Present() before the screenshot)
// Writes a BMP screenshot (what SDL_SaveBMP(screen, "screenshot.bmp") does at :175/:147) and, optionally, the
// raw XRGB words so tests can compare them with the oracle.
#include "mirt_draw.hpp"

#include <cstdlib>
#include <iostream>

using namespace mirt_host;

int main(int argc, char **argv)
{
    const std::string which = argc > 1 ? argv[1] : "rt";
    const int W = argc > 3 ? std::atoi(argv[2]) : 500, H = argc > 3 ? std::atoi(argv[3]) : 500;
    const char *bmp = argc > 4 ? argv[4] : "screenshot.bmp";
    const char *raw = argc > 5 ? argv[5] : nullptr;
    try {
        std::vector<uint32_t> pixels((size_t)W * H, 0u), back((size_t)W * H, 0u);
        Surface screen = { pixels.data(), W, H, W * 4 };            // InitializeSDL(W, H): 32-bit SWSURFACE
        Surface second = { back.data(), W, H, W * 4 };              // *async: the surface being drawn while `screen` is shown
        const bool async = which == "rtasync" || which == "rasterasync";
        // the loop of the *async modes: eight frames with the camera turning back to where it started, surfaces in turn, so
        // that the last frame -- the reference view -- lands in `screen`
        auto async_loop = [&](auto &app) {
            check(mirt_surface_register(screen.pixels, (size_t)H * screen.pitch), "mirt_surface_register");
            check(mirt_surface_register(second.pixels, (size_t)H * second.pitch), "mirt_surface_register");
            check(mirt_set_frames_in_flight(2), "mirt_set_frames_in_flight");
            for (int loop = 0; loop < 8; loop++) {
                app.yaw = 0.05f * (float)(7 - loop);
                app.Update();
                app.DrawAsync((loop & 1) ? screen : second);
            }
            app.Present();
            check(mirt_surface_unregister(second.pixels), "mirt_surface_unregister");
            check(mirt_surface_unregister(screen.pixels), "mirt_surface_unregister");
        };
        if (which == "rt" || which == "rtsoft" || which == "rtaa" || which == "rtdof" || which == "rtasync") {
            RayTracer app;
            app.SOFT_SHADOWS_ENABLED = which == "rtsoft";
            app.AA_ENABLED = which == "rtaa";
            app.DOF_ENABLED = which == "rtdof";
            app.SCREEN_WIDTH = W; app.SCREEN_HEIGHT = H;
            app.focalLength = (float)H / 2.0f;                       // 250 for the reference's 500x500
            app.screen = screen;
            app.AddLight(vec3(0, -0.5f, -0.7f), vec3(1, 1, 1), 14);  // raytracer.cpp:116 (draws the soft-shadow jitter from
                                                                     // rand(): do it before anything else can touch the stream)
            check(mirt_init(0), "mirt_init");
            check(mirt_set_profiling(1), "mirt_set_profiling");       // "Render time" below is the GPU time of the call
            app.LoadTestModel();                                     // :149
            app.cameraRot[1][1] = 1.0f;                              // :162
            if (async) async_loop(app);
            else for (int loop = 0; loop < 2; loop++) {              // while (NoQuitMessageSDL())
                app.Update();
                if (app.isUpdated) { app.Draw(); app.isUpdated = false; }
            }
        } else {
            Rasteriser app;
            app.DOF_ENABLED = which == "rasterdof";
            app.SCREEN_WIDTH = W; app.SCREEN_HEIGHT = H;
            app.focalLength = (float)H;                              // 500 for the reference's 500x500
            app.screen = screen;
            app.AddLight(vec3(0, -0.5f, -0.7f), vec3(1, 1, 1), 14);  // rasteriser.cpp:104
            check(mirt_init(0), "mirt_init");
            check(mirt_set_profiling(1), "mirt_set_profiling");
            if (which == "rasterstl") {                              // #ifdef CUSTOM_MODEL (:106-110)
                app.LoadSTLFile(argc > 6 ? argv[6] : "Source/enemy1.stl");
                app.cameraPos = vec3(0, -0.5f, -5.0f);
                app.GPU_CULL = true;
            } else {
                app.LoadTestModel();                                 // :112
            }
            app.cameraRot[1][1] = 1.01f;                             // :115 (sic)
            if (async) async_loop(app);
            else for (int loop = 0; loop < 2; loop++) {
                app.Update();
                if (app.isUpdated) { app.Draw(); app.isUpdated = false; }
            }
        }
        mirt_stats st;
        check(mirt_get_stats(&st), "mirt_get_stats");
        std::cout << "Render time: " << st.gpu_ms << " ms." << std::endl;   // the reference's only metric (:343)
        save_bmp(screen, bmp);
        if (raw) { FILE *f = std::fopen(raw, "wb"); if (f) { std::fwrite(pixels.data(), 4, pixels.size(), f); std::fclose(f); } }
        mirt_shutdown();
    } catch (const std::exception &e) {
        std::cerr << "demo_main: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
