"""tools/fuzz_binned.py [first_seed count [frames]] -- (GPU box) binned vs brute-force frames on many seeded random configurations
(scene size and triangle size, cameras inside / outside, up to four lights some of them grazing a triangle's plane or
sitting on a vertex, soft shadows, supersampling, bands).  Prints every mismatch; exit code 1 if there was one."""
import sys

import numpy as np

sys.path.insert(0, "cpp-raytracer-rasterizer_amd")
import mirt

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
# frames per configuration: the first four of a light position bin its cube themselves, the fifth builds the shared fine cube,
# the sixth reads it from the cache -- 6 covers all three paths (1: the first only)
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 1
mirt.init(0)
bad = 0
for seed in range(first, first + count):
    rng = np.random.RandomState(seed)
    n = int(rng.choice([520, 900, 2500, 7000, 20000]))
    size = float(rng.choice([0.01, 0.04, 0.1, 0.3, 0.9]))
    tris = mirt.scene_soup(seed, n, size)
    if rng.rand() < 0.3:
        tris = np.concatenate([tris, mirt.scene_cornell()])
    W, H = int(rng.randint(65, 500)), int(rng.randint(65, 400))
    cam = rng.uniform(-1.3, 1.3, 3) if rng.rand() < 0.4 else np.array([rng.uniform(-0.7, 0.7), rng.uniform(-0.7, 0.7), -rng.uniform(1.2, 4.0)])
    yaw = float(rng.uniform(-3.14, 3.14))
    rot = np.zeros(9, np.float32); rot[0] = np.cos(np.float32(yaw)); rot[2] = -np.sin(np.float32(yaw)); rot[6] = np.sin(np.float32(yaw)); rot[8] = np.cos(np.float32(yaw)); rot[4] = 1.0
    rot[2], rot[6] = rot[6], rot[2]          # cameraRot[0][2] = sin, cameraRot[2][0] = -sin (column-major)
    focal = float(rng.uniform(0.2, 1.5) * H)
    nl = int(rng.randint(1, 5))
    lights = np.zeros((nl, 7), np.float32)
    lights[:, 0:3] = rng.uniform(-1.1, 1.1, (nl, 3))
    lights[:, 3:6] = rng.uniform(0.1, 1.0, (nl, 3))
    lights[:, 6] = rng.uniform(2, 25, nl)
    mode = rng.randint(4)
    k = int(rng.randint(len(tris)))
    v0, e1, e2, nrm = tris[k, 0:3], tris[k, 3:6] - tris[k, 0:3], tris[k, 6:9] - tris[k, 0:3], tris[k, 9:12]
    if mode == 1:
        lights[0, 0:3] = v0 + rng.uniform(-3, 3) * e1 + rng.uniform(-3, 3) * e2                       # in a triangle's plane
    elif mode == 2:
        lights[0, 0:3] = v0 + 0.3 * e1 + 0.3 * e2 + np.float32(rng.choice([1e-6, 1e-4, 1e-2])) * nrm   # just above a triangle
    elif mode == 3:
        lights[0, 0:3] = v0                                                                          # on a vertex
    samples = 1
    if rng.rand() < 0.15 and nl <= 2:
        samples = 4
        jit = (np.repeat(lights[:, 0:3], samples, axis=0) + (rng.rand(nl * samples, 3).astype(np.float32) - np.float32(0.5)) * np.float32(0.08)).astype(np.float32)
        mirt.set_soft_shadows(samples, jit)
    aa = 2 if rng.rand() < 0.1 else 1
    mirt.set_antialiasing(aa)
    try:
        mirt.scene_upload(tris)
        view = mirt.make_view(cam, rot, focal, W, H)
        b = mirt.raytrace(view, lights, mode=mirt.RT_BRUTE)
        for rep in range(frames):
            a = mirt.raytrace(view, lights, mode=mirt.RT_BINNED)
            if not (np.array_equal(a["index"], b["index"]) and np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32)) and np.array_equal(a["xrgb"], b["xrgb"])):
                break
    finally:
        mirt.set_soft_shadows(1)
        mirt.set_antialiasing(1)
    ok = np.array_equal(a["index"], b["index"]) and np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32)) and np.array_equal(a["xrgb"], b["xrgb"]) and a["stats"]["shadow_rays"] == b["stats"]["shadow_rays"]
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, "n", len(tris), "size", size, "WxH", W, H, "lights", nl, "mode", mode, "samples", samples, "aa", aa,
              "index diffs", int((a["index"] != b["index"]).sum()), "word diffs", int((a["xrgb"] != b["xrgb"]).sum()), flush=True)
print("fuzz: %d configurations from seed %d, %d frame(s) each, %d mismatches" % (count, first, frames, bad))
sys.exit(1 if bad else 0)
