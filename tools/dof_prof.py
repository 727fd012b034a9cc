#!/usr/bin/env python3
"""tools/dof_prof.py [libmirt variant.so] -- rasterised Cornell frames at 4K with the 8x8 depth-of-field blur (static camera): per-kernel
times, and a target for rocprofv3 --pmc (A/B runs of dof_kernel.hip variants on the GPU box)."""
import sys

import numpy as np

sys.path[:0] = ["cpp-raytracer-rasterizer_amd", "tests"]
import mirt                                 # noqa: E402
if len(sys.argv) > 1:
    mirt.LIB_PATH = sys.argv[1]
from devbuf import DeviceArray              # noqa: E402

W, H = 3840, 2160
LIGHT = np.array([[0.0, -0.5, -0.7, 1, 1, 1, 14]], np.float32)
mirt.init(0)
tris = mirt.scene_cornell()
view = mirt.make_view((0, 0, -3), mirt.rot_from_yaw(0.0, 1.01), 2160.0, W, H)
mirt.scene_upload(tris, mirt.cull(tris, view, 3))
mirt.set_depth_of_field(8, 1.9)
x = DeviceArray((H, W), np.uint32)
mirt.set_profiling(True)
acc = {}
for it in range(16):
    mirt.rasterise_device(view, LIGHT, (0.2, 0.2, 0.2), 0, H, 0, x.ptr, W * 4)
    mirt.sync()
    st = mirt.stats()
    if it >= 6:
        for k, v in st["kernel_ms"].items():
            acc[k] = acc.get(k, 0.0) + v / 10
print("kernel_ms %s" % {k: round(v, 4) for k, v in acc.items() if v})
mirt.shutdown()
