"""tools/threshold_sweep.py -- (GPU box) where should MIRT_RT_AUTO start binning?  Frame time of brute force (LDS-resident /
chunked kernels) against binned mode for mid-size soups at 1080p, two frames in flight."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "cpp-raytracer-rasterizer_amd")
import mirt

mirt.init(0)
W, H = 1920, 1080
rot = np.zeros(9, np.float32); rot[0] = rot[4] = rot[8] = 1
view = mirt.make_view((0, 0, -2), rot, 540.0, W, H)
light = np.array([[0, -0.5, -0.7, 1, 1, 1, 14]], np.float32)
bufs = [torch.zeros((H, W), dtype=torch.int32, device="cuda") for _ in range(2)]
mirt.set_frames_in_flight(2)
for n in (65, 100, 150, 200, 300, 400, 511, 800):
    tris = mirt.scene_soup(3, n, 0.25)
    mirt.scene_upload(tris)
    row = []
    for mode in (mirt.RT_BRUTE, mirt.RT_BINNED):
        calls = [mirt.prepared_raytrace_device(view, light, (0.2, 0.2, 0.2), mode, 0, H, 0, b.data_ptr(), W * 4) for b in bufs]
        for i in range(6):
            calls[i & 1]()
        mirt.sync()
        t0 = time.perf_counter()
        reps = 40
        for i in range(reps):
            calls[i & 1]()
        mirt.sync()
        row.append((time.perf_counter() - t0) / reps * 1e3)
    print("n=%4d  brute %.4f ms  binned %.4f ms" % (n, row[0], row[1]), flush=True)
