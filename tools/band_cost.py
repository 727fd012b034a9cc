#!/usr/bin/env python3
"""tools/band_cost.py [world] -- what each rank of a `world`-way band split renders and bins on BASELINE config 5 (1 M triangles,
7680 x 4320): every band rendered on this ONE GPU in turn (binned path, one frame in flight), per-kernel GPU time of the band's
frame next to the whole frame's.  The numbers DESIGN.md section 7 quotes for the multi-GPU split."""
import sys

import numpy as np

sys.path[:0] = ["cpp-raytracer-rasterizer_amd", "tests"]
import mirt                                 # noqa: E402
from devbuf import DeviceArray              # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W, H = 7680, 4320
LIGHT = np.array([[0.0, -0.5, -0.7, 1, 1, 1, 14]], np.float32)
mirt.init(0)
mirt.scene_upload(mirt.scene_soup(2, 1000000, 0.02))
view = mirt.make_view((0, 0, -2), mirt.rot_from_yaw(0.0, 1.0), H / 2.0, W, H)
x = DeviceArray((H, W), np.uint32)
mirt.set_profiling(True)


for it in range(8):                          # (the light settles into the shared cube on the way: its one-off build stays out of the numbers)
    mirt.raytrace_device(view, LIGHT, (0.2, 0.2, 0.2), mirt.RT_BINNED, 0, H, 0, x.ptr, W * 4)
mirt.sync()


def cost(y0, y1):
    acc = {}
    for it in range(8):
        mirt.raytrace_device(view, LIGHT, (0.2, 0.2, 0.2), mirt.RT_BINNED, y0, y1, 0, x.ptr, W * 4)
        mirt.sync()
        st = mirt.stats()
        if it >= 3:
            for k, v in st["kernel_ms"].items():
                acc[k] = acc.get(k, 0.0) + v / 5
    return {k: round(v, 4) for k, v in acc.items() if v}, st["primary_rays"] + st["shadow_rays"]


full, rays = cost(0, H)
print("whole frame: kernel_ms %s, %d rays" % (full, rays))
for r in range(world):
    y0, y1 = mirt.band_of(r, world, H)
    c, rays = cost(y0, y1)
    print("band %d rows [%d, %d): kernel_ms %s, %d rays" % (r, y0, y1, c, rays))
mirt.shutdown()
