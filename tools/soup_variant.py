#!/usr/bin/env python3
"""tools/soup_variant.py [libmirt variant.so] -- trace / bin kernel time of the 100 k soup at 1080p with an alternative build
of the library (A/B runs of kernel variants on the GPU box)."""
import sys

import numpy as np

sys.path[:0] = ["cpp-raytracer-rasterizer_amd", "tests"]
import mirt                                 # noqa: E402
if len(sys.argv) > 1:
    mirt.LIB_PATH = sys.argv[1]
from devbuf import DeviceArray              # noqa: E402

LIGHT = np.array([[0.0, -0.5, -0.7, 1, 1, 1, 14]], np.float32)
W, H = 1920, 1080
mirt.init(0)
mirt.scene_upload(mirt.scene_soup(1, 100000, 0.05))
view = mirt.make_view((0, 0, -2), mirt.rot_from_yaw(0.0, 1.0), 540.0, W, H)
x = DeviceArray((H, W), np.uint32)
mirt.set_profiling(True)
for lights, tag in ((LIGHT, "1 light"), (np.zeros((0, 7), np.float32), "no light")):
    acc = {}
    for it in range(16):
        mirt.raytrace_device(view, lights, (0.2, 0.2, 0.2), mirt.RT_BINNED, 0, H, 0, x.ptr, W * 4)
        mirt.sync()
        st = mirt.stats()
        if it >= 6:
            for k, v in st["kernel_ms"].items():
                acc[k] = acc.get(k, 0.0) + v / 10
    print("%-9s kernel_ms %s tests %d candidates %d steps p/s %d/%d drains %d" % (tag, {k: round(v, 4) for k, v in acc.items() if v}, st["tests"],
          st["candidates"], st["steps_primary"], st["steps_shadow"], st["drains"]))
mirt.shutdown()
