"""tools/host_path_rate.py -- (GPU box) frame rate through the host-buffer entry points (the boundary the reference's Draw()
binds: the surface is host memory), i.e. including the device-to-host copy of the XRGB frame.  bench.py's `value` is the
device-resident rate; this is the PCIe-inclusive figure DESIGN.md quotes next to it."""
import sys
import time

import numpy as np

sys.path.insert(0, "cpp-raytracer-rasterizer_amd")
import mirt

mirt.init(0)
rot = np.zeros(9, np.float32); rot[0] = rot[4] = rot[8] = 1
light = np.array([[0, -0.5, -0.7, 1, 1, 1, 14]], np.float32)
tris = mirt.scene_cornell()
mirt.scene_upload(tris)
for (W, H) in ((500, 500), (1920, 1080)):
    view = mirt.make_view((0, 0, -2), rot, H / 2.0, W, H)
    surf = np.zeros((H, W), np.uint32)
    for _ in range(5):
        mirt.raytrace(view, light, want_rgb=False, want_index=False, xrgb=surf)
    t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        mirt.raytrace(view, light, want_rgb=False, want_index=False, xrgb=surf)
    dt = (time.perf_counter() - t0) / n
    print("ray tracer %dx%d through mirt_raytrace (host surface): %.3f ms/frame, %.0f frames/s, %.1f MB/frame D2H" % (W, H, dt * 1e3, 1 / dt, W * H * 4 / 1e6))
rot[4] = 1.01
W, H = 3840, 2160
view = mirt.make_view((0, 0, -3), rot, float(H), W, H)
mirt.scene_upload(tris, mirt.cull(tris, view, 3))
for _ in range(3):
    mirt.rasterise(view, light, want_rgb=False, want_zinv=False, want_index=False)
t0 = time.perf_counter()
n = 30
for _ in range(n):
    mirt.rasterise(view, light, want_rgb=False, want_zinv=False, want_index=False)
dt = (time.perf_counter() - t0) / n
print("rasteriser %dx%d through mirt_rasterise (host surface, fresh numpy array per call): %.3f ms/frame, %.0f frames/s, %.1f MB/frame D2H" % (W, H, dt * 1e3, 1 / dt, W * H * 4 / 1e6))
