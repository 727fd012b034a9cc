"""tools/moving_light.py -- (GPU box) the 100 k soup at 1080p, binned, with the LIGHT moving every frame (the reference moves it
with the W/A/S/D/Q/E keys, raytracer.cpp:152-162): every frame rebuilds the light-cube bins and the expanded light rows that
a moving camera alone reuses.  Prints ms per frame for: static everything, moving camera, moving light, both."""
import sys
import time

import numpy as np

sys.path[:0] = ["cpp-raytracer-rasterizer_amd", "tests"]
import mirt
from devbuf import DeviceArray

mirt.init(0)
W, H = 1920, 1080
tris = mirt.scene_soup(1, 100000, 0.05)
mirt.scene_upload(tris)
outs = [DeviceArray((H, W), np.uint32, 0), DeviceArray((H, W), np.uint32, 0)]
ind = (0.5, 0.5, 0.5)
views = [mirt.make_view((0, 0, -2.0), mirt.rot_from_yaw(0.001 * i, 1.0), float(H), W, H) for i in range(64)]
lights = [np.array([[0.0 + 0.001 * i, -0.5, -0.7, 1, 1, 1, 14]], np.float32) for i in range(64)]


def run(move_cam, move_light, frames=300):
    for i in range(8):
        mirt.raytrace_device(views[i % 64 if move_cam else 0], lights[i % 64 if move_light else 0], ind, mirt.RT_BINNED, 0, H, 0, outs[i & 1].ptr, W * 4)
    mirt.sync()
    t0 = time.perf_counter()
    for i in range(frames):
        mirt.raytrace_device(views[i % 64 if move_cam else 0], lights[i % 64 if move_light else 0], ind, mirt.RT_BINNED, 0, H, 0, outs[i & 1].ptr, W * 4)
    mirt.sync()
    return (time.perf_counter() - t0) / frames * 1e3


cases = ((False, False), (True, False), (False, True), (True, True))
if len(sys.argv) > 1:                        # "light": only the moving-light case (for a profiler run)
    cases = ((False, True),)
for in_flight in (1, 2):
    mirt.set_frames_in_flight(in_flight)
    for mc, ml in cases:
        print("%d in flight: camera %-6s light %-6s  %.4f ms per frame" % (in_flight, "moves" if mc else "fixed", "moves" if ml else "fixed", run(mc, ml)), flush=True)
mirt.shutdown()
