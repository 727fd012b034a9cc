"""tools/fuzz_raster_sequence.py [first_seed count] -- (GPU box) random call sequences through the rasteriser's device entry points:
scenes of 30 .. 6000 triangles (both sides of the 4096-triangle limit of the worst-case span table), the cull step on the device
before every frame (as Update() does, rasteriser.cpp:404-447) with changing flags, cameras that move or stand still, one or two
frames in flight, whole frames and bands -- against the CPU restatement (oracle/mirt_oracle.c): XRGB words of every frame.
Uses the oracle: a tool for the GPU box's test side, like tests/."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "cpp-raytracer-rasterizer_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import mirt                                  # noqa: E402
from devbuf import DeviceArray               # noqa: E402
from mirt_oracle import Oracle               # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 30
o = Oracle()
mirt.init(0)
bad = 0
IND = (0.2, 0.2, 0.2)
for seed in range(first, first + count):
    if (seed - first) % 50 == 49:
        print("... %d sequences, %d mismatching frames so far" % (seed - first + 1, bad), flush=True)
    rng = np.random.RandomState((15485863 * seed + 11) % (1 << 32))
    W, H = int(rng.choice([120, 200, 257])), int(rng.choice([90, 131, 160]))
    outs = [DeviceArray((H, W), np.uint32, 0x5A) for _ in range(3)]
    scratch = DeviceArray((H, W), np.uint32, 0x5A)
    k = 0
    mirt.set_frames_in_flight(1)
    in_flight = 1
    scene = None
    for step in range(int(rng.randint(8, 16))):
        if scene is None or rng.rand() < 0.2:
            mirt.sync()
            n = int(rng.choice([30, 200, 1500, 4096, 4097, 6000]))
            scene = mirt.scene_cornell() if n == 30 else mirt.scene_soup(int(rng.randint(1 << 30)), n, float(rng.choice([0.05, 0.15, 0.4])))
            mirt.scene_upload(scene)
            cur_culled = np.zeros(len(scene), np.uint8)
        if rng.rand() < 0.15:
            mirt.sync()
            in_flight = int(rng.choice([1, 2, 3, 4]))
            mirt.set_frames_in_flight(in_flight)
        nl = int(rng.randint(1, 3))
        L = np.zeros((nl, 7), np.float32)
        L[:, 0:3] = rng.uniform(-0.9, 0.9, (nl, 3)); L[:, 3:6] = rng.uniform(0.3, 1.0, (nl, 3)); L[:, 6] = rng.uniform(4, 18, nl)
        run = []
        for j in range(int(rng.randint(1, 4))):
            cam = (float(rng.uniform(-0.4, 0.4)), float(rng.uniform(-0.4, 0.4)), float(rng.uniform(-3.2, -1.0)))
            rot = o.rot_from_yaw(float(rng.uniform(-0.5, 0.5)), 1.01)
            focal = float(rng.uniform(0.6, 1.2)) * H
            # the cull step is optional: a frame drawn without it uses the flags of the most recent one (of the upload: none culled)
            flags = int(rng.randint(4)) if rng.rand() < 0.7 else -1
            if flags >= 0:
                cur_culled = o.cull(scene, cam, rot, focal, W, H, flags)
            want = o.rasterise(scene, cur_culled, cam, rot, focal, W, H, L)["xrgb"]
            run.append((mirt.make_view(cam, rot, focal, W, H), flags, want))
        pending = []
        for (v, flags, want) in run:
            bi = k % 3; k += 1
            banded = rng.rand() < 0.25
            if flags >= 0:
                mirt.cull_device(v, flags)                  # the flags of the NEXT call (with several frames in flight: of its stream)
            if banded:
                ys = int(rng.randint(1, H - 1))
                mirt.rasterise_device(v, L, IND, 0, ys, 0, outs[bi].ptr, W * 4)
                mirt.rasterise_device(v, L, IND, ys, H, 0, outs[bi].ptr, W * 4)   # (the second call lands on another stream and fetches the flags)
            else:
                if rng.rand() < 0.3:                        # a ray-traced frame takes the turn the cull step expected: the flags must follow the rasteriser
                    mirt.raytrace_device(v, L, IND, mirt.RT_AUTO, 0, H, 0, scratch.ptr, W * 4)
                mirt.rasterise_device(v, L, IND, 0, H, 0, outs[bi].ptr, W * 4)
            pending.append((bi, want, "step %d n %d flags %d in_flight %d banded %s" % (step, len(scene), flags, in_flight, banded)))
        mirt.sync()
        for (bi, want, tag) in pending:
            got = outs[bi].read()
            if not np.array_equal(got, want):
                bad += 1
                ys, xs = np.nonzero(got != want)
                print("MISMATCH seed", seed, tag, "words", int((got != want).sum()), "first at (x, y, got, want):",
                      [(int(x), int(y), hex(int(got[y, x])), hex(int(want[y, x]))) for y, x in list(zip(ys, xs))[:6]], flush=True)
    mirt.set_frames_in_flight(1)
    for b in outs + [scratch]:
        b.free()
print("fuzz: %d rasteriser call sequences from seed %d against the oracle, %d mismatching frames" % (count, first, bad))
mirt.shutdown()
sys.exit(1 if bad else 0)
