"""tools/tile_split.py -- (GPU box) k_rt_tile2 on the Cornell box at 1080p with no light / one light / two lights: how the
kernel's time splits between the primary pass (ray setup, masks, closest hit) and DirectLight (per light: shading + shadow ray)."""
import sys

import numpy as np

sys.path[:0] = ["cpp-raytracer-rasterizer_amd", "tests"]
import mirt
from devbuf import DeviceArray

W, H = 1920, 1080
mirt.init(0)
mirt.scene_upload(mirt.scene_cornell())
view = mirt.make_view((0, 0, -2), mirt.rot_from_yaw(0.0, 1.0), 540.0, W, H)
x = DeviceArray((H, W), np.uint32)
mirt.set_profiling(True)
L2 = np.array([[0.0, -0.5, -0.7, 1, 1, 1, 14], [0.3, -0.4, -0.2, 1, 1, 1, 5]], np.float32)
for lights, tag in ((L2[:0], "no light"), (L2[:1], "1 light"), (L2, "2 lights")):
    acc = 0.0
    for it in range(14):
        mirt.raytrace_device(view, lights, (0.2, 0.2, 0.2), mirt.RT_AUTO, 0, H, 0, x.ptr, W * 4)
        mirt.sync()
        st = mirt.stats()
        if it >= 4:
            acc += st["kernel_ms"]["trace"] / 10
    print("%-9s trace %.4f ms  tests %d" % (tag, acc, st["tests"]))
mirt.shutdown()
