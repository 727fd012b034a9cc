"""tools/fuzz_small.py [first_seed count] -- (GPU box) small random scenes (3..64 triangles: the tile kernels; some larger ones:
the LDS-resident and chunked brute-force kernels) through the ray tracer -- plain, 16-sample soft shadows, 2x2 supersampling, one
to three lights, ragged frame sizes -- and through the rasteriser, against the CPU restatement (oracle/mirt_oracle.c): index,
float colour bits, depth bits and XRGB words.  Prints every mismatch; exit code 1 if there was one.  Uses the oracle: a tool
for the GPU box's test side, like tests/ (never part of the product path)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "cpp-raytracer-rasterizer_amd"), os.path.join(ROOT, "oracle")]
import mirt                                  # noqa: E402
if os.environ.get("MIRT_LIB"):                # A/B runs of a library variant
    mirt.LIB_PATH = os.environ["MIRT_LIB"]
from mirt_oracle import Oracle               # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 500
o = Oracle()
mirt.init(0)
bad = 0
for seed in range(first, first + count):
    rng = np.random.RandomState((7919 * seed + 13) % (1 << 32))
    n = int(rng.choice([3, 5, 9, 17, 30, 31, 33, 63, 64, 65, 200, 900]))
    size = float(rng.uniform(0.2, 1.6))
    tris = mirt.scene_soup(seed, n, size)
    if rng.rand() < 0.3:
        tris = np.concatenate([mirt.scene_cornell(), tris])[:max(n, 30)]
    W, H = int(rng.choice([33, 64, 70, 127, 200, 257])), int(rng.choice([17, 40, 64, 90, 131]))
    yaw = float(rng.uniform(-0.6, 0.6))
    cam = (float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-3.0, -1.2)))
    focal = float(rng.uniform(0.4, 1.2)) * H
    nl = int(rng.randint(1, 4))
    lights = np.zeros((nl, 7), np.float32)
    lights[:, 0:3] = rng.uniform(-0.9, 0.9, (nl, 3))
    lights[:, 3:6] = rng.uniform(0.2, 1.0, (nl, 3))
    lights[:, 6] = rng.uniform(3, 20, nl)
    variant = int(rng.randint(4))                        # 0 plain, 1 soft shadows, 2 supersampling, 3 rasteriser
    rot_o = o.rot_from_yaw(yaw, 1.01 if variant == 3 else 1.0)      # ONE matrix for both sides (cosf and numpy's cos differ in the last bit)
    view = mirt.make_view(cam, rot_o, focal, W, H)
    what = ""
    try:
        if variant == 3:
            flags = int(rng.randint(4))
            culled = o.cull(tris, cam, rot_o, focal, W, H, flags)
            mirt.scene_upload(tris, culled)
            g = mirt.rasterise(view, lights)
            r = o.rasterise(tris, culled, cam, rot_o, focal, W, H, lights)
            ok = np.array_equal(g["xrgb"], r["xrgb"]) and np.array_equal(g["rgb"].view(np.uint32), r["rgb"].view(np.uint32)) and \
                np.array_equal(g["depth"].view(np.uint32), r["depth"].view(np.uint32))
            what = "raster flags %d | words %d rgb %d depth %d" % (flags, int((g["xrgb"] != r["xrgb"]).sum()), int((g["rgb"].view(np.uint32) != r["rgb"].view(np.uint32)).sum()),
                                                                   int((g["depth"].view(np.uint32) != r["depth"].view(np.uint32)).sum()))
        else:
            samples, aa, jit = 1, 1, None
            if variant == 1:
                samples = 16 if nl == 1 else 4
                jit = (np.repeat(lights[:, 0:3], samples, axis=0) + (rng.rand(nl * samples, 3).astype(np.float32) - np.float32(0.5)) * np.float32(0.08)).astype(np.float32)
                mirt.set_soft_shadows(samples, jit)
            if variant == 2:
                aa = 2
                mirt.set_antialiasing(aa)
            mirt.scene_upload(tris)
            g = mirt.raytrace(view, lights)
            r = o.raytrace(tris, cam, rot_o, focal, W, H, lights, samples=samples, jitter=jit, aa=aa, threads=8)
            inner = (slice(1, H - 1), slice(1, W - 1))
            ok = np.array_equal(g["index"], r["index"]) and np.array_equal(g["rgb"].view(np.uint32), r["rgb"].view(np.uint32)) and \
                np.array_equal(g["xrgb"][inner], r["xrgb"][inner])
            what = "rt samples %d aa %d mode %d | index %d rgb %d words %d" % (samples, aa, g["stats"]["mode_used"], int((g["index"] != r["index"]).sum()),
                                                                              int((g["rgb"].view(np.uint32) != r["rgb"].view(np.uint32)).sum()), int((g["xrgb"][inner] != r["xrgb"][inner]).sum()))
    finally:
        mirt.set_soft_shadows(1)
        mirt.set_antialiasing(1)
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, "n", len(tris), "WxH", W, H, "lights", nl, what, flush=True)
print("fuzz: %d small configurations from seed %d through the ray tracer and the rasteriser against the oracle, %d mismatches" % (count, first, bad))
mirt.shutdown()
sys.exit(1 if bad else 0)
