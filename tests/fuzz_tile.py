"""tests/fuzz_tile.py [first_seed count] -- (GPU box) the tile ray tracer (scenes of at most 64 triangles, two pixels per
lane) and the rasteriser against the CPU oracle on many seeded random small configurations.  Test infrastructure (it loads
the oracle, so it lives under tests/); not collected by pytest -- run it by hand from the repository root."""
import sys

import numpy as np

sys.path.insert(0, "cpp-raytracer-rasterizer_amd")
sys.path.insert(0, "oracle")
import mirt
from mirt_oracle import Oracle

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
o = Oracle()
mirt.init(0)
bad = 0
for seed in range(first, first + count):
    rng = np.random.RandomState(seed)
    n = int(rng.choice([1, 3, 12, 30, 50, 64]))
    tris = mirt.scene_soup(seed, n, float(rng.choice([0.2, 0.6, 1.2, 2.5])))
    if rng.rand() < 0.3:
        tris = mirt.scene_cornell() if n >= 30 else tris
    W, H = int(rng.randint(3, 180)), int(rng.randint(3, 140))
    cam = np.array([rng.uniform(-0.8, 0.8), rng.uniform(-0.8, 0.8), -rng.uniform(0.2, 3.5)], np.float32)
    rot = o.rot_from_yaw(float(rng.uniform(-1.2, 1.2)), 1.0)
    focal = float(rng.uniform(0.3, 1.3) * H)
    nl = int(rng.randint(0, 4))
    lights = np.zeros((nl, 7), np.float32)
    if nl:
        lights[:, 0:3] = rng.uniform(-1.1, 1.1, (nl, 3)); lights[:, 3:6] = rng.uniform(0.1, 1.0, (nl, 3)); lights[:, 6] = rng.uniform(2, 25, nl)
    aa = int(rng.choice([1, 1, 1, 2, 3]))
    mirt.scene_upload(tris)
    mirt.set_antialiasing(aa)
    try:
        got = mirt.raytrace(mirt.make_view(cam, rot, focal, W, H), lights, mode=mirt.RT_AUTO)
    finally:
        mirt.set_antialiasing(1)
    ref = o.raytrace(tris, cam, rot, focal, W, H, lights, threads=8, aa=aa)
    ok = np.array_equal(got["index"], ref["index"]) and np.array_equal(got["rgb"].view(np.uint32), ref["rgb"].view(np.uint32)) and np.array_equal(got["xrgb"], ref["xrgb"])
    # the same scene through the rasteriser
    rotr = o.rot_from_yaw(float(rng.uniform(-1.2, 1.2)), 1.01)
    view = mirt.make_view(cam, rotr, focal, W, H)
    flags = int(rng.randint(4))
    culled = mirt.cull(tris, view, flags)
    okr = np.array_equal(culled, o.cull(tris, cam, rotr, focal, W, H, flags))
    mirt.scene_upload(tris, culled)
    gr = mirt.rasterise(view, lights)
    rr = o.rasterise(tris, culled, cam, rotr, focal, W, H, lights)
    okr = okr and np.array_equal(gr["index"], rr["index"]) and np.array_equal(gr["rgb"].view(np.uint32), rr["rgb"].view(np.uint32)) and np.array_equal(gr["xrgb"], rr["xrgb"])
    if not (ok and okr):
        bad += 1
        print("MISMATCH seed", seed, "n", len(tris), "WxH", W, H, "lights", nl, "aa", aa, "rt ok", ok, "raster ok", okr, flush=True)
print("fuzz: %d small configurations from seed %d through the ray tracer and the rasteriser, %d mismatches" % (count, first, bad))
sys.exit(1 if bad else 0)
