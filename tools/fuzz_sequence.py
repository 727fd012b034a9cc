"""tools/fuzz_sequence.py [first_seed count] -- (GPU box) random SEQUENCES of calls through the device entry points: scenes come and
go (30 .. 30 000 triangles), lights move, stand still, change in number, the camera moves or stands still, frames are queued on one
or two streams, bands and whole frames alternate, soft shadows / supersampling switch on and off -- the state machine around the
kernels (light-cube cache and per-frame light pass, guessed list sizes, cull flags per stream, scratch growth).  Every frame of a
sequence is compared with the brute-force frame of its own parameters.  Prints every mismatch; exit code 1 if there was one."""
import sys

import numpy as np

sys.path[:0] = ["cpp-raytracer-rasterizer_amd", "tests"]
import mirt
from devbuf import DeviceArray

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
mirt.init(0)
bad = 0
IND = (0.2, 0.2, 0.2)
for seed in range(first, first + count):
    if (seed - first) % 50 == 49:
        print("... %d sequences, %d mismatching frames so far" % (seed - first + 1, bad), flush=True)
    rng = np.random.RandomState((104729 * seed + 7) % (1 << 32))
    W, H = int(rng.choice([160, 256, 333])), int(rng.choice([96, 144, 200]))
    outs = [DeviceArray((H, W), np.uint32, 0x5A) for _ in range(4)]
    ref = DeviceArray((H, W), np.uint32, 0x5A)
    scene, lights, view, aa, soft = None, None, None, 1, None
    pending = []                                            # (buffer index, expected image)
    in_flight = 1
    mirt.set_frames_in_flight(1)

    def new_scene():
        n = int(rng.choice([30, 300, 2500, 9000, 30000]))
        t = mirt.scene_soup(int(rng.randint(1 << 30)), n, float(rng.choice([0.03, 0.08, 0.25])))
        if n == 30:
            t = mirt.scene_cornell()
        mirt.scene_upload(t)
        return t

    def new_lights(old):
        nl = int(rng.randint(1, 4)) if old is None or rng.rand() < 0.3 else len(old)
        L = np.zeros((nl, 7), np.float32)
        L[:, 0:3] = rng.uniform(-0.9, 0.9, (nl, 3)); L[:, 3:6] = rng.uniform(0.3, 1.0, (nl, 3)); L[:, 6] = rng.uniform(4, 18, nl)
        return L

    def new_view():
        return mirt.make_view((float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-2.8, -1.4))),
                              mirt.rot_from_yaw(float(rng.uniform(-0.4, 0.4)), 1.0), float(rng.uniform(0.5, 1.1)) * H, W, H)

    def drain():
        global bad
        mirt.sync()
        for (bi, want, tag) in pending:
            got = outs[bi].read()
            if not np.array_equal(got, want):
                bad += 1
                print("MISMATCH seed", seed, tag, "words", int((got != want).sum()), flush=True)
        pending.clear()

    scene, lights, view = new_scene(), new_lights(None), new_view()
    k = 0
    for step in range(int(rng.randint(25, 45))):
        r = rng.rand()
        if r < 0.06:
            drain(); scene = new_scene()
        elif r < 0.30:
            lights = new_lights(lights)
        elif r < 0.36:
            if rng.rand() < 0.5:
                lights = lights.copy(); lights[0, 0] += np.float32(0.01)
        if rng.rand() < 0.7:
            view = new_view()
        if rng.rand() < 0.08:
            drain()
            aa = int(rng.choice([1, 1, 2]))
            mirt.set_antialiasing(aa)
        if rng.rand() < 0.08:
            drain()
            if rng.rand() < 0.5 and len(lights) <= 2:
                samples = 4
                soft = (np.repeat(lights[:, 0:3], samples, axis=0) + (rng.rand(len(lights) * samples, 3).astype(np.float32) - np.float32(0.5)) * np.float32(0.08)).astype(np.float32)
                mirt.set_soft_shadows(samples, soft)
            else:
                soft = None
                mirt.set_soft_shadows(1)
        if soft is not None and len(soft) != 4 * len(lights):
            drain(); soft = None; mirt.set_soft_shadows(1)
        if rng.rand() < 0.1:
            drain()
            in_flight = int(rng.choice([1, 2, 3, 4]))
            mirt.set_frames_in_flight(in_flight)
        # a run of up to four frames with these settings (camera and/or first light moving from frame to frame): the expected
        # frames first (brute force, one at a time), then the binned frames queued back to back -- overlapping when two are in flight
        run = []
        for j in range(int(rng.randint(1, 5))):
            if j and rng.rand() < 0.7:
                view = new_view()
            if j and rng.rand() < 0.4:
                lights = lights.copy(); lights[0, 1] += np.float32(0.02)
                if soft is not None:
                    soft = soft.copy(); soft[0:4, 1] += np.float32(0.02); mirt.set_soft_shadows(4, soft)
            mirt.raytrace_device(view, lights, IND, mirt.RT_BRUTE, 0, H, 0, ref.ptr, W * 4)
            run.append((view, lights, None if soft is None else soft.copy(), ref.read()))
        mode = mirt.RT_BINNED if len(scene) > 64 else mirt.RT_AUTO
        for (v, L, sft, want) in run:
            if sft is not None:
                mirt.sync(); mirt.set_soft_shadows(4, sft)
            bi = k % 4; k += 1
            if rng.rand() < 0.25:                           # the frame as two bands
                ys = int(rng.randint(1, H - 1))
                mirt.raytrace_device(v, L, IND, mode, 0, ys, 0, outs[bi].ptr, W * 4)
                mirt.raytrace_device(v, L, IND, mode, ys, H, 0, outs[bi].ptr, W * 4)
            else:
                mirt.raytrace_device(v, L, IND, mode, 0, H, 0, outs[bi].ptr, W * 4)
            pending.append((bi, want, "step %d n %d lights %d aa %d soft %s in_flight %d" % (step, len(scene), len(L), aa, sft is not None, in_flight)))
        drain()
    drain()
    mirt.set_antialiasing(1); mirt.set_soft_shadows(1); mirt.set_frames_in_flight(1)
    for o in outs + [ref]:
        o.free()
print("fuzz: %d call sequences from seed %d, %d mismatching frames" % (count, first, bad))
mirt.shutdown()
sys.exit(1 if bad else 0)
