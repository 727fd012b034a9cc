#!/usr/bin/env python3
"""tools/trace_prof.py [nolight|1m8k|cornell] [variant.so] -- a dozen binned frames of the 100 k soup at 1080p (static camera), for rocprofv3:
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES ... -d gpurun_out/x -- python3 tools/trace_prof.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "cpp-raytracer-rasterizer_amd"), os.path.join(ROOT, "tests")]
import mirt                                 # noqa: E402
for a in sys.argv[1:]:
    if a.endswith(".so"):
        mirt.LIB_PATH = a            # an A/B build (tools/build_variant.sh)
from devbuf import DeviceArray              # noqa: E402

big = "1m8k" in sys.argv                    # BASELINE config 5 (1 M triangles, 8K) instead of config 3 (100 k, 1080p)
W, H = (7680, 4320) if big else (1920, 1080)
lights = np.zeros((0, 7), np.float32) if "nolight" in sys.argv else np.array([[0.0, -0.5, -0.7, 1, 1, 1, 14]], np.float32)
mirt.init(0)
cornell = "cornell" in sys.argv              # BASELINE config 2: the Cornell box at 1080p (tile kernel)
mirt.scene_upload(mirt.scene_cornell() if cornell else mirt.scene_soup(2, 1000000, 0.02) if big else mirt.scene_soup(1, 100000, 0.05))
view = mirt.make_view((0, 0, -3), mirt.rot_from_yaw(0.0, 1.0), float(H), W, H) if cornell else mirt.make_view((0, 0, -2), mirt.rot_from_yaw(0.0, 1.0), H / 2.0, W, H)
x = DeviceArray((H, W), np.uint32)
move = "move" in sys.argv                   # every frame its own yaw (and a sync behind it): the chain of a single frame, camera moving
for it in range(12):
    if move:
        view = mirt.make_view((0, 0, -2), mirt.rot_from_yaw(0.001 * it, 1.0), H / 2.0, W, H)
    mirt.raytrace_device(view, lights, (0.2, 0.2, 0.2), mirt.RT_AUTO if cornell else mirt.RT_BINNED, 0, H, 0, x.ptr, W * 4)
    if move:
        mirt.sync()
mirt.sync()
print(mirt.stats())
mirt.shutdown()
