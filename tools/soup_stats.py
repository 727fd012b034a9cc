#!/usr/bin/env python3
"""Where the binned ray-trace frame of the 100 k soup spends its tests (GPU box): kernel time with and without the light,
tests per primary / shadow ray, and how coherent the shadow rays of an 8x8 tile are (distinct light-cube bins per tile)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "cpp-raytracer-rasterizer_amd"), os.path.join(ROOT, "tests")]
import mirt                                 # noqa: E402
from devbuf import DeviceArray              # noqa: E402

LIGHT = np.array([[0.0, -0.5, -0.7, 1, 1, 1, 14]], np.float32)
W, H = 1920, 1080
mirt.init(0)
tris = mirt.scene_soup(1, 100000, 0.05)
mirt.scene_upload(tris)
rot = mirt.rot_from_yaw(0.0, 1.0)
view = mirt.make_view((0, 0, -2), rot, 540.0, W, H)
x, idx, pos = DeviceArray((H, W), np.uint32), DeviceArray((H, W), np.int32), DeviceArray((H, W, 3), np.float32)
dist = DeviceArray((H, W), np.float32)
mirt.set_profiling(True)
for lights, tag in ((LIGHT, "1 light"), (np.zeros((0, 7), np.float32), "no light")):
    acc = {}
    for it in range(12):
        mirt.raytrace_device(view, lights, (0.2, 0.2, 0.2), mirt.RT_BINNED, 0, H, 0, x.ptr, W * 4, None, idx.ptr, dist.ptr, pos.ptr)
        mirt.sync()
        st = mirt.stats()
        if it >= 2:
            for k, v in st["kernel_ms"].items():
                acc[k] = acc.get(k, 0.0) + v / 10
    print("%-9s kernel_ms %s  primary %d shadow %d tests %d" % (tag, {k: round(v, 4) for k, v in acc.items() if v}, st["primary_rays"], st["shadow_rays"], st["tests"]))
mirt.raytrace_device(view, LIGHT, (0.2, 0.2, 0.2), mirt.RT_BINNED, 0, H, 0, x.ptr, W * 4, None, idx.ptr, dist.ptr, pos.ptr)
index, P = idx.read(), pos.read()
hit = index >= 0
print("hit pixels %d of %d" % (hit.sum(), W * H))
# light-cube bin of every hit pixel (cube_bin_of, rt_binned.hpp; 64 x 64 bins per face)
L = LIGHT[0, :3]
d = (L - P).astype(np.float32)
d /= np.maximum(np.linalg.norm(d, axis=2, keepdims=True), 1e-30)
a = np.abs(d)
k = np.argmax(a, axis=2)
m = np.take_along_axis(a, k[..., None], 2)[..., 0]
sg = np.take_along_axis(d, k[..., None], 2)[..., 0]
u = np.take_along_axis(d, ((k + 1) % 3)[..., None], 2)[..., 0] / m
v = np.take_along_axis(d, ((k + 2) % 3)[..., None], 2)[..., 0] / m
B = 64
i = np.clip(np.floor((u + 1) * B / 2), 0, B - 1).astype(np.int64)
j = np.clip(np.floor((v + 1) * B / 2), 0, B - 1).astype(np.int64)
bins = ((2 * k + (sg < 0)) * B + j) * B + i
bins[~hit] = -1
for T in (8, 16, 32):
    th, tw = H // T, W // T
    b = bins[:th * T, :tw * T].reshape(th, T, tw, T).transpose(0, 2, 1, 3).reshape(th * tw, T * T)
    distinct = np.array([len(np.unique(r[r >= 0])) for r in b])
    nh = (b >= 0).sum(axis=1)
    live = nh > 0
    print("tile %2d: tiles with hits %d, hits/tile %.1f, distinct bins/tile mean %.1f median %d p90 %d max %d, rays per (tile,bin) %.2f"
          % (T, live.sum(), nh[live].mean(), distinct[live].mean(), np.median(distinct[live]), np.percentile(distinct[live], 90), distinct.max(), nh[live].sum() / distinct[live].sum()))
ub, cnt = np.unique(bins[hit], return_counts=True)
print("global: %d distinct bins hold the %d shadow rays: mean %.1f rays per bin, median %d, max %d; bins with >= 64 rays hold %.1f%% of the rays"
      % (len(ub), hit.sum(), cnt.mean(), np.median(cnt), cnt.max(), 100.0 * cnt[cnt >= 64].sum() / hit.sum()))
D = dist.read()
print("hit distance: mean %.3f, p5 %.3f, p95 %.3f" % (D[hit].mean(), np.percentile(D[hit], 5), np.percentile(D[hit], 95)))
mirt.shutdown()
