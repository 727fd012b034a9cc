#!/usr/bin/env python3
"""tools/moving_ab.py <soup100k|soup1m8k> [NAME=VALUE ...] -- (GPU box) frame time of the binned ray tracer with four frames and with one frame in flight and the
camera MOVING (yaw += 1 mrad per frame, 64 views, as bench.py does: every frame runs its binning pass), once with the environment as it
is and once per NAME=VALUE (or alternative libmirt .so) given (a child process each: the library reads its switches when it initialises)."""
import os
import subprocess
import sys
import time

import numpy as np

if len(sys.argv) > 2 and sys.argv[1] == "--child":
    sys.path[:0] = ["cpp-raytracer-rasterizer_amd", "tests"]
    import mirt
    from devbuf import DeviceArray
    work, tag = sys.argv[2], sys.argv[3]
    if tag.endswith(".so"):
        mirt.LIB_PATH = tag                 # an A/B build (tools/build_variant.sh)
    LIGHT = np.array([[0.0, -0.5, -0.7, 1, 1, 1, 14]], np.float32)
    IND = (0.2, 0.2, 0.2)
    mirt.init(0)
    if work == "soup1m8k":
        W, H, n, size, seed, reps, frames = 7680, 4320, 1000000, 0.02, 2, 3, 200
    else:
        W, H, n, size, seed, reps, frames = 1920, 1080, 100000, 0.05, 1, 5, 2000
    mirt.scene_upload(mirt.scene_soup(seed, n, size))
    views = [mirt.make_view((0, 0, -2), mirt.rot_from_yaw(0.001 * i, 1.0), H / 2.0, W, H) for i in range(64)]
    flights = [int(x) for x in os.environ.get("MIRT_AB_FLIGHT", "4,1").split(",")]
    bufs = [DeviceArray((H, W), np.uint32) for _ in range(max(flights))]
    res = []
    for in_flight in flights:
        mirt.set_frames_in_flight(in_flight)
        best = 1e9
        for rep in range(reps):
            for i in range(frames // 10):
                mirt.raytrace_device(views[i & 63], LIGHT, IND, mirt.RT_BINNED, 0, H, 0, bufs[i % in_flight].ptr, W * 4)
            mirt.sync()
            t0 = time.perf_counter()
            for i in range(frames):
                mirt.raytrace_device(views[i & 63], LIGHT, IND, mirt.RT_BINNED, 0, H, 0, bufs[i % in_flight].ptr, W * 4)
            mirt.sync()
            best = min(best, (time.perf_counter() - t0) / frames)
        res.append(best * 1e6)
    print("%-10s %-40s frame %s us with %s in flight (camera moving, best of %d x %d)" % (work, tag.split("/")[-1], " / ".join("%.2f" % r for r in res), " / ".join(str(f) for f in flights), reps, frames), flush=True)
    mirt.shutdown()
    sys.exit(0)

work = sys.argv[1]
for setting in ["-"] + sys.argv[2:]:
    env = dict(os.environ)
    if setting != "-" and not setting.endswith(".so"):
        k, v = setting.split("=", 1)
        env[k] = v
    subprocess.run([sys.executable, sys.argv[0], "--child", work, setting], env=env, check=False)
