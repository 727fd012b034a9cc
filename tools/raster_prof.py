#!/usr/bin/env python3
"""tools/raster_prof.py -- a dozen rasterised frames of the Cornell box at 4K (static camera), for rocprofv3:
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES ... -d gpurun_out/x -- python3 tools/raster_prof.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "cpp-raytracer-rasterizer_amd"), os.path.join(ROOT, "tests")]
import mirt                                 # noqa: E402
if len(sys.argv) > 1:
    mirt.LIB_PATH = sys.argv[1]             # an alternative build of the library (tools/build_variant.sh)
from devbuf import DeviceArray              # noqa: E402

W, H = 3840, 2160
LIGHT = np.array([[0.0, -0.5, -0.7, 1, 1, 1, 14]], np.float32)
mirt.init(0)
tris = mirt.scene_cornell()
view = mirt.make_view((0, 0, -3), mirt.rot_from_yaw(0.0, 1.01), 2160.0, W, H)
mirt.scene_upload(tris, mirt.cull(tris, view, 3))
x = DeviceArray((H, W), np.uint32)
mirt.set_profiling(True)
for it in range(12):
    mirt.rasterise_device(view, LIGHT, (0.2, 0.2, 0.2), 0, H, 0, x.ptr, W * 4)
    mirt.sync()
print(mirt.stats())
mirt.shutdown()
