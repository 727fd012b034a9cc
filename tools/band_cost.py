#!/usr/bin/env python3
"""tools/band_cost.py [world] [equal|weighted|both] -- what each rank of a `world`-way band split renders and bins on BASELINE config 5
(1 M triangles, 7680 x 4320): every band rendered on this ONE GPU in turn (binned path, one frame in flight, the view moving from
frame to frame as in bench.py so that every frame runs its binning pass), per-kernel GPU time of the band's frame next to the whole
frame's.  `weighted`: the bands mirt_weighted_bounds derives from the whole frame's cost histogram (MIRT_PARTITION_WEIGHTED), which is
what every rank of a sharded run computes for itself.  The numbers DESIGN.md section 7 quotes for the multi-GPU split."""
import sys

import numpy as np

sys.path[:0] = ["cpp-raytracer-rasterizer_amd", "tests"]
import mirt                                 # noqa: E402
from devbuf import DeviceArray              # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
which = sys.argv[2] if len(sys.argv) > 2 else "both"
W, H = 7680, 4320
LIGHT = np.array([[0.0, -0.5, -0.7, 1, 1, 1, 14]], np.float32)
mirt.init(0)
mirt.scene_upload(mirt.scene_soup(2, 1000000, 0.02))
views = [mirt.make_view((0, 0, -2), mirt.rot_from_yaw(0.001 * i, 1.0), H / 2.0, W, H) for i in range(8)]
x = DeviceArray((H, W), np.uint32)
mirt.set_profiling(True)
mirt.set_cost_histogram(True)

for it in range(8):                          # (the light settles into the shared cube on the way: its one-off build stays out of the numbers)
    mirt.raytrace_device(views[it % 8], LIGHT, (0.2, 0.2, 0.2), mirt.RT_BINNED, 0, H, 0, x.ptr, W * 4)
mirt.sync()


def cost(y0, y1):
    acc = {}
    for it in range(10):
        mirt.raytrace_device(views[it % 8], LIGHT, (0.2, 0.2, 0.2), mirt.RT_BINNED, y0, y1, 0, x.ptr, W * 4)
        mirt.sync()
        st = mirt.stats()
        if it >= 4:
            for k, v in st["kernel_ms"].items():
                acc[k] = acc.get(k, 0.0) + v / 6
    return {k: round(v, 4) for k, v in acc.items() if v}, st


full, st = cost(0, H)
print("whole frame: kernel_ms %s, %d rays, %d candidates, %d triangles selected" % (full, st["primary_rays"] + st["shadow_rays"], st["candidates"], st["selected_triangles"]))
hist, shift = mirt.cost_histogram()
print("cost histogram: %d coarse rows of %d tile rows, sum %d" % (0 if hist is None else len(hist), 1 << shift, 0 if hist is None else int(hist.sum())))
for name in (("equal", "weighted") if which == "both" else (which,)):
    bounds = [mirt.band_of(r, world, H)[0] for r in range(world)] + [H] if name == "equal" else mirt.weighted_bounds(hist, shift, W, H, world)
    print("%s bands: %s" % (name, bounds))
    tot, bins = [], []
    for r in range(world):
        c, st = cost(bounds[r], bounds[r + 1])
        tot.append(c.get("bin", 0.0) + c.get("trace", 0.0)); bins.append(c.get("bin", 0.0))
        print("  band %d rows [%d, %d): kernel_ms %s, %d rays, %d candidates, %d triangles selected" % (r, bounds[r], bounds[r + 1], c, st["primary_rays"] + st["shadow_rays"], st["candidates"], st["selected_triangles"]))
    print("  %s: sum of bin %.4f ms (%.2f x the whole frame's), slowest band bin + trace %.4f ms, mean %.4f, slowest / mean %.3f"
          % (name, sum(bins), sum(bins) / max(full.get("bin", 1e-9), 1e-9), max(tot), sum(tot) / len(tot), max(tot) / (sum(tot) / len(tot))))
mirt.shutdown()
