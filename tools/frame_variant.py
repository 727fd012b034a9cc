#!/usr/bin/env python3
"""tools/frame_variant.py <cornell1080|raster4k|soup100k> [libmirt variant.so ...] -- (GPU box) frame time with four frames in flight and the
kernels' times alone, for the shipped library and for alternative builds of it (tools/build_variant.sh): A/B runs of kernel variants.
One child process per library (the library is loaded once per process)."""
import subprocess
import sys
import time

import numpy as np

if len(sys.argv) > 2 and sys.argv[1] == "--child":
    sys.path[:0] = ["cpp-raytracer-rasterizer_amd", "tests"]
    import mirt
    work, lib = sys.argv[2], sys.argv[3]
    if lib != "-":
        mirt.LIB_PATH = lib
    from devbuf import DeviceArray
    LIGHT = np.array([[0.0, -0.5, -0.7, 1, 1, 1, 14]], np.float32)
    IND = (0.2, 0.2, 0.2)
    mirt.init(0)
    if work == "raster4k":
        W, H = 3840, 2160
        tris = mirt.scene_cornell()
        view = mirt.make_view((0, 0, -3), mirt.rot_from_yaw(0.0, 1.01), 2160.0, W, H)
        mirt.scene_upload(tris, mirt.cull(tris, view, 3))
        draw = lambda p: mirt.rasterise_device(view, LIGHT, IND, 0, H, 0, p, W * 4)
    elif work == "soup100k":
        W, H = 1920, 1080
        mirt.scene_upload(mirt.scene_soup(1, 100000, 0.05))
        view = mirt.make_view((0, 0, -2), mirt.rot_from_yaw(0.0, 1.0), 540.0, W, H)
        draw = lambda p: mirt.raytrace_device(view, LIGHT, IND, mirt.RT_BINNED, 0, H, 0, p, W * 4)
    else:
        W, H = 1920, 1080
        mirt.scene_upload(mirt.scene_cornell())
        view = mirt.make_view((0, 0, -3), mirt.rot_from_yaw(0.0, 1.0), 1080.0, W, H)
        draw = lambda p: mirt.raytrace_device(view, LIGHT, IND, mirt.RT_AUTO, 0, H, 0, p, W * 4)
    bufs = [DeviceArray((H, W), np.uint32) for _ in range(4)]
    mirt.set_profiling(True)
    acc = {}
    for it in range(16):
        draw(bufs[0].ptr)
        mirt.sync()
        if it >= 6:
            for k, v in mirt.stats()["kernel_ms"].items():
                acc[k] = acc.get(k, 0.0) + v / 10
    mirt.set_profiling(False)
    mirt.set_frames_in_flight(4)
    best = 1e9
    for rep in range(5):
        for i in range(200):
            draw(bufs[i & 3].ptr)
        mirt.sync()
        t0 = time.perf_counter()
        for i in range(2000):
            draw(bufs[i & 3].ptr)
        mirt.sync()
        best = min(best, (time.perf_counter() - t0) / 2000)
    mirt.set_frames_in_flight(1)
    one = 1e9
    for rep in range(3):
        for i in range(100):
            draw(bufs[0].ptr)
        mirt.sync()
        t0 = time.perf_counter()
        for i in range(1000):
            draw(bufs[0].ptr)
        mirt.sync()
        one = min(one, (time.perf_counter() - t0) / 1000)
    print("%-44s frame %.2f us (4 in flight, best of 5 x 2000), %.2f us with one in flight   alone: %s" % (lib.split("/")[-1], best * 1e6, one * 1e6, {k: round(v * 1e3, 1) for k, v in acc.items() if v}), flush=True)
    mirt.shutdown()
    sys.exit(0)

work = sys.argv[1]
for lib in ["-"] + sys.argv[2:]:
    subprocess.run([sys.executable, sys.argv[0], "--child", work, lib], timeout=120)
