#!/usr/bin/env python3
"""tools/bin_stamps.py <libmirt built with -DMIRT_BIN_STAMPS> [soup100k|soup1m8k] -- (GPU box) where the workgroups of k_bin_pairs spend their clock
ticks (set-up / prefix / flattened rounds / huge items / flush; s_memtime ticks of 10 ns): two frames of the moving camera, every 97th workgroup prints."""
import sys

import numpy as np

sys.path[:0] = ["cpp-raytracer-rasterizer_amd", "tests"]
import mirt                                 # noqa: E402
mirt.LIB_PATH = sys.argv[1]
from devbuf import DeviceArray              # noqa: E402

work = sys.argv[2] if len(sys.argv) > 2 else "soup100k"
W, H, n, size, seed = (7680, 4320, 1000000, 0.02, 2) if work == "soup1m8k" else (1920, 1080, 100000, 0.05, 1)
LIGHT = np.array([[0.0, -0.5, -0.7, 1, 1, 1, 14]], np.float32)
mirt.init(0)
mirt.scene_upload(mirt.scene_soup(seed, n, size))
x = DeviceArray((H, W), np.uint32)
for it in range(4):
    if it == 3:
        print("---- frame %d (camera pass only) ----" % it, flush=True)
    view = mirt.make_view((0, 0, -2), mirt.rot_from_yaw(0.001 * it, 1.0), H / 2.0, W, H)
    mirt.raytrace_device(view, LIGHT, (0.2, 0.2, 0.2), mirt.RT_BINNED, 0, H, 0, x.ptr, W * 4)
    mirt.sync()
mirt.shutdown()
