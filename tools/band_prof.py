#!/usr/bin/env python3
"""tools/band_prof.py y0 y1 [frames] [move] -- BASELINE config 5 (1 M triangles, 7680 x 4320), rows [y0, y1) only, binned path, one frame
in flight; for rocprofv3 (per-kernel time of ONE band's frame: what a rank of the band split runs):
    rocprofv3 --kernel-trace --stats -d gpurun_out/x -- python3 tools/band_prof.py 0 540
`move`: every frame gets its own yaw (1 mrad apart), as bench.py's moving camera does; otherwise the view stands still."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "cpp-raytracer-rasterizer_amd"), os.path.join(ROOT, "tests")]
import mirt                                 # noqa: E402
for a in sys.argv[1:]:
    if a.endswith(".so"):
        mirt.LIB_PATH = a                   # an A/B build (tools/build_variant.sh)
from devbuf import DeviceArray              # noqa: E402

W, H = 7680, 4320
y0, y1 = int(sys.argv[1]), int(sys.argv[2])
frames = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 16
move = "move" in sys.argv
LIGHT = np.array([[0.0, -0.5, -0.7, 1, 1, 1, 14]], np.float32)
mirt.init(0)
mirt.scene_upload(mirt.scene_soup(2, 1000000, 0.02))
x = DeviceArray((H, W), np.uint32)
mirt.set_profiling(True)
acc = {}
for it in range(8 + frames):
    view = mirt.make_view((0, 0, -2), mirt.rot_from_yaw(0.001 * (it % 64) if move else 0.0, 1.0), H / 2.0, W, H)
    mirt.raytrace_device(view, LIGHT, (0.2, 0.2, 0.2), mirt.RT_BINNED, y0, y1, 0, x.ptr, W * 4)
    mirt.sync()
    if it >= 8:
        for k, v in mirt.stats()["kernel_ms"].items():
            acc[k] = acc.get(k, 0.0) + v / frames
print("rows [%d, %d): kernel_ms %s" % (y0, y1, {k: round(v, 4) for k, v in acc.items() if v}))
mirt.shutdown()
