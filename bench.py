#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native render path (see BASELINE.json / SURVEY section 8(d)).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A "step" is one frame of the hot path over a synthetic scene whose inputs are already resident in HBM.
Prints ONE JSON line (rank 0).  Workloads (BASELINE.json `configs`):
    cornell1080  (default, configs[1]) Cornell box, 1920x1080, primary + DirectLight shadow rays
    soup100k     (configs[2]) 100k random triangles, 1920x1080
    raster4k     (configs[3]) rasteriser, Cornell box, 3840x2160
    cornell500   (configs[0]) the reference's own 500x500 case
    cornell1080soft16  configs[1] with the reference's 16-sample soft shadows switched on
    cornell1080aa3     configs[1] with AA_SAMPLES = 3 supersampling
    cornell1080dof8, raster4kdof8   configs[1] / configs[3] with the 8x8 depth-of-field blur
    soup1m8k     (configs[4]) 1M random triangles, 7680x4320 (meant for 8 GPUs)
With --gpus N > 1 (launched by torch.distributed.run, one rank per GPU) the frame is split into N bands of
rows; every rank renders its band and the XRGB bands are gathered on rank 0 over RCCL ("scaling": "strong"); frames
that render in microseconds travel 32 to a gather.  MIRT_BENCH_REHEARSAL=1 runs that control flow with every rank on
device 0 (gloo, host-staged gathers) and checks the assembled frames against a single-GPU frame -- not a measurement.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cpp-raytracer-rasterizer_amd"))

import numpy as np  # noqa: E402

LIGHT = np.array([[0.0, -0.5, -0.7, 1.0, 1.0, 1.0, 14.0]], np.float32)     # raytracer.cpp:116 / rasteriser.cpp:104
INDIRECT = (0.2, 0.2, 0.2)                                                    # raytracer.cpp:81

FLOP_PER_TEST = 60.0          # 57 add/mul + 3 div as written in raytracer.cpp:216-239
PEAK_FP32_TFLOPS = 157.3      # MI355X_MICROARCH.md: vector FP32 peak (counts FMA as 2; this path may not fuse)
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec peak

WORKLOADS = {
    #  name         kind      scene                      W     H     cam            focal   rot11
    "cornell1080": ("rt", ("cornell",), 1920, 1080, (0, 0, -2), 540.0, 1.0),
    "cornell500": ("rt", ("cornell",), 500, 500, (0, 0, -2), 250.0, 1.0),
    "cornell1080soft16": ("rt", ("cornell",), 1920, 1080, (0, 0, -2), 540.0, 1.0),   # + SOFT_SHADOWS_SAMPLES = 16
    "cornell1080aa3": ("rt", ("cornell",), 1920, 1080, (0, 0, -2), 540.0, 1.0),      # + AA_SAMPLES = 3 (rank 2)
    "cornell1080dof8": ("rt", ("cornell",), 1920, 1080, (0, 0, -2), 540.0, 1.0),     # + DOF_KERNEL_SIZE = 8, FOCAL_LENGTH = 1.3 (rank 3)
    "raster4kdof8": ("raster", ("cornell",), 3840, 2160, (0, 0, -3), 2160.0, 1.01),  # + DOF_KERNEL_SIZE = 8, FOCAL_LENGTH = 1.9
    "soup100k": ("rt", ("soup", 1, 100000, 0.05), 1920, 1080, (0, 0, -2), 540.0, 1.0),
    "soup1m8k": ("rt", ("soup", 2, 1000000, 0.02), 7680, 4320, (0, 0, -2), 2160.0, 1.0),
    "raster4k": ("raster", ("cornell",), 3840, 2160, (0, 0, -3), 2160.0, 1.01),
}


def identity_rot(m11):
    r = np.zeros(9, np.float32)
    r[0] = r[8] = 1.0          # yaw = 0: cos = 1, sin = 0 (raytracer.cpp:377-382)
    r[4] = m11
    return r


def measured_traffic(workload, kernel_prefixes):
    """HBM bytes per launch of the named kernel(s) from the committed rocprofv3 PMC summary (profiles/
    r01_hbm_traffic.json: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 FETCH correction applied).  bench.py
    cannot run the profiler on itself, so `traffic` is the last profiled value for this exact workload, or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")) as f:
            w = json.load(f)["workloads"].get(workload, {})
    except (OSError, ValueError):
        return None
    tot = 0
    for k, v in w.items():
        if any(k.replace("mirt::", "").startswith(p) for p in kernel_prefixes):
            tot += v["fetch_bytes"] + v["write_bytes"]
    return tot or None


def measured_valu_instructions(workload, kernel_prefix):
    """Wave-level VALU instructions per launch of the named kernel from the committed rocprofv3 PMC pass
    (profiles/r01_pmc_issue_<workload>.txt, SQ_INSTS_VALU averaged over the launches), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_issue_%s.txt" % workload)) as f:
            for line in f:
                if line.startswith("pmc1") and kernel_prefix in line and "SQ_INSTS_VALU" in line:
                    return float(line.split("'SQ_INSTS_VALU':")[1].split(",")[0].strip(" }\n"))
    except (OSError, ValueError, IndexError):
        pass
    return None


def cpu_baseline(kind, tris, culled, W, H, cam, rot, focal, budget_s=12.0, samples=1, jitter=None, aa=1):
    """The oracle (CPU restatement, oracle/mirt_oracle.c) timed on this host's cores on a bounded sample of
    the same workload.  Test infrastructure: measured as a baseline, never used by the product path."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from mirt_oracle import Oracle
    o = Oracle()
    cores = len(os.sched_getaffinity(0))
    if kind == "raster":
        t0 = time.perf_counter()
        reps = 0
        while True:
            o.rasterise(tris, culled, cam, rot, focal, W, H, LIGHT, want=())
            reps += 1
            if time.perf_counter() - t0 > min(budget_s, 6.0) or reps >= 5:
                break
        dt = (time.perf_counter() - t0) / reps
        return {"value": round(1.0 / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                "sample": "%d full %dx%d frames, single thread (the reference's default for the rasteriser)" % (reps, W, H)}
    # ray tracer: time a probe of rows first, then as many evenly spaced rows as fit the budget (bounded sample)
    cores = min(cores, 64)                      # OpenMP over rows stops scaling long before 256 threads here
    centre = H // 2
    t0 = time.perf_counter()
    soft = dict(samples=samples, jitter=jitter, aa=aa)
    r = o.raytrace(tris, cam, rot, focal, W, H, LIGHT, y0=centre, y1=centre + 1, threads=cores, want=("xrgb",), **soft)
    probe = max(time.perf_counter() - t0, 1e-4)
    rows = int(max(cores, min(H, budget_s / probe)))
    if rows >= H:
        t0 = time.perf_counter()
        r = o.raytrace(tris, cam, rot, focal, W, H, LIGHT, threads=cores, want=("xrgb",), **soft)
        dt = time.perf_counter() - t0
        rays = W * H * aa * aa + r["nshadow"]
        sample = "full %dx%d frame" % (W, H)
    else:
        # one call over a contiguous band of `rows` rows around the image centre keeps all threads busy
        ya = max(0, centre - rows // 2)
        yb = min(H, ya + rows)
        t0 = time.perf_counter()
        r = o.raytrace(tris, cam, rot, focal, W, H, LIGHT, y0=ya, y1=yb, threads=cores, want=("xrgb",), **soft)
        dt = time.perf_counter() - t0
        rays = W * (yb - ya) * aa * aa + r["nshadow"]
        sample = "rows %d..%d of %d (central band), per-ray rate" % (ya, yb - 1, H)
    return {"value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port", "sample": sample}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="cornell1080", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="auto", choices=["auto", "brute", "binned"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world

    import torch
    import torch.distributed as dist
    import mirt

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the render path has no CPU fallback")
    # MIRT_BENCH_REHEARSAL=1: every rank on device 0 with a gloo group and host-staged gathers -- the N > 1 control flow
    # (batches, events, reductions) on a one-GPU box.  Not a measurement.
    rehearsal = world > 1 and os.environ.get("MIRT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    mirt.init(local_rank)

    kind, scene, W, H, cam, focal, rot11 = WORKLOADS[args.workload]
    rot = identity_rot(rot11)
    tris = mirt.scene_cornell() if scene[0] == "cornell" else mirt.scene_soup(scene[1], scene[2], scene[3])
    view = mirt.make_view(cam, rot, focal, W, H)
    culled = mirt.cull(tris, view, 3) if kind == "raster" else None
    mirt.scene_upload(tris, culled)
    mode = {"auto": mirt.RT_AUTO, "brute": mirt.RT_BRUTE, "binned": mirt.RT_BINNED}[args.mode]
    soft_samples, soft_jitter = 1, None
    if args.workload.endswith("soft16"):
        # AddLight's jitter (raytracer.cpp:186-190): 16 positions, each coordinate light + U[-0.04, 0.04] (synthetic here:
        # a fixed-seed numpy stream instead of the C library's rand())
        soft_samples = 16
        soft_jitter = (LIGHT[:, 0:3] + (np.random.RandomState(1).rand(16, 3).astype(np.float32) - np.float32(0.5)) * np.float32(0.08)).astype(np.float32)
        mirt.set_soft_shadows(soft_samples, soft_jitter)
    aa = 3 if args.workload.endswith("aa3") else 1
    mirt.set_antialiasing(aa)
    dof = 8 if args.workload.endswith("dof8") else 0
    mirt.set_depth_of_field(dof, 1.3 if kind == "rt" else 1.9)

    steps = args.steps if args.steps is not None else (1000 if len(tris) < 1000 else 20)
    warmup = args.warmup if args.warmup is not None else (50 if len(tris) < 1000 else 3)

    from mirt.sharding import BandGather
    dev = torch.device("cuda", local_rank)
    depth = 2
    # Several GPUs: frames that render faster than a collective starts (the 30-triangle scenes: ~25 us per frame, a band of
    # it a few us) travel `batch` at a time -- one RCCL gather moves the bands of 32 consecutive frames; heavy frames (the
    # soups: milliseconds) go one per gather.  Every frame is still rendered, gathered and assembled inside the timed region.
    batch = 32 if (world > 1 and len(tris) < 1000) else 1
    bands = BandGather(H, W, dev, depth=depth, batch=batch, via_host=rehearsal)  # this rank's two band buffers (+ the gathered frames on rank 0)
    y0, y1 = bands.y0, bands.y1
    # One GPU: two frames in flight, alternating between the two bands (the library alternates between two streams, so
    # the next frame is dispatched while the previous one drains).  Several GPUs: one frame in flight per rank, the
    # RCCL gather of the previous batch overlaps it instead.
    mirt.set_frames_in_flight(2 if world == 1 else 1)
    mirt_stream = torch.cuda.ExternalStream(mirt.load().mirt_stream(), device=dev)
    renders = []                                   # renders[k][b]: enqueue one frame into slot b of band buffer k
    for k in range(depth):
        row = []
        for b in range(batch):
            ptr = bands.slot(k, b).data_ptr()
            if kind == "rt":
                row.append(mirt.prepared_raytrace_device(view, LIGHT, INDIRECT, mode, y0, y1, y0, ptr, W * 4))
            else:
                row.append(mirt.prepared_rasterise_device(view, LIGHT, INDIRECT, y0, y1, y0, ptr, W * 4))
        renders.append(row)
    render = renders[0][0]
    comm_stream = torch.cuda.Stream(device=dev) if world > 1 else None
    rendered = [torch.cuda.Event() for _ in range(depth)]      # band buffer k holds a finished batch
    gathered = [torch.cuda.Event() for _ in range(depth)]      # band buffer k has been sent and may be overwritten
    frame_no = [0]

    def flush(k):
        """The batch in band buffer k is complete: gather it on the communication stream."""
        rendered[k].record(mirt_stream)
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(rendered[k])
            bands.gather(k)
            gathered[k].record(comm_stream)

    def step():
        i = frame_no[0]
        frame_no[0] += 1
        if world == 1:
            renders[i & 1][0]()
            return
        # Double-buffered band buffers: batch j renders into buffer j%2 on mirt's stream while the RCCL gather of batch
        # j-1 (the other buffer) is still in flight on the communication stream.  Dependencies are two events per buffer.
        b, k = i % batch, (i // batch) % depth
        if b == 0:
            mirt_stream.wait_event(gathered[k])                # the gather that last read buffer k has finished
        renders[k][b]()
        if b == batch - 1:
            flush(k)

    last_batch = [batch]                             # frames the most recent gather carried

    def finish_batch():
        """Gathers a batch the loop left incomplete and restarts the batch numbering."""
        last_batch[0] = batch
        if world > 1 and frame_no[0] % batch:
            last_batch[0] = frame_no[0] % batch
            flush((frame_no[0] // batch) % depth)
        frame_no[0] = 0

    def fence():
        mirt.sync()
        if comm_stream is not None:
            comm_stream.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    finish_batch()
    fence()
    st = mirt.stats()
    t0 = time.perf_counter()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if world > 1:
        ev0.record(mirt_stream)
    for _ in range(steps):
        step()
    finish_batch()
    if world > 1:
        ev1.record(mirt_stream)
    fence()
    dt = time.perf_counter() - t0
    # HIP events on the stream the kernels run on (one stream per rank when sharded; with two frames in flight the
    # frames alternate between two streams and the wall clock between the fences is the measure)
    gpu_total_ms = ev0.elapsed_time(ev1) if world > 1 else dt * 1e3

    if rehearsal and rank == 0:
        # the frames rank 0 assembled from every rank's bands against one full-frame render (interior words; the border
        # is never written by the ray tracer)
        full = torch.zeros((H, W), dtype=torch.int32, device=dev)
        if kind == "rt":
            mirt.raytrace_device(view, LIGHT, INDIRECT, mode, 0, H, 0, full.data_ptr(), W * 4)
        else:
            mirt.rasterise_device(view, LIGHT, INDIRECT, 0, H, 0, full.data_ptr(), W * 4)
        mirt.sync()
        got = bands.frames[: last_batch[0]] if batch > 1 else bands.frame.unsqueeze(0)
        same = all(bool(torch.equal(got[b], full)) for b in range(got.shape[0]))
        print("rehearsal: %d gathered frame(s) identical to the single-GPU frame: %s" % (got.shape[0], same), file=sys.stderr, flush=True)
        if not same:
            raise SystemExit("rehearsal: gathered frames differ from the single-GPU frame")

    # whole-job numbers: MAX over ranks of the wall time, SUM over ranks of the rays
    st = mirt.stats()
    rays_rank = float(st["primary_rays"] + st["shadow_rays"]) if kind == "rt" else 0.0
    tests_rank = float(st["tests"])
    if world > 1:
        rdev = torch.device("cpu") if rehearsal else dev
        t = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        s = torch.tensor([rays_rank, tests_rank, float(st["shadow_rays"])], dtype=torch.float64, device=rdev)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        rays_frame, tests_frame, shadow_frame = [float(x) for x in s.tolist()]
    else:
        rays_frame, tests_frame, shadow_frame = rays_rank, tests_rank, float(st["shadow_rays"])

    # per-kernel durations of this rank: a separate profiled pass (events around every launch)
    mirt.set_frames_in_flight(1)
    mirt.set_profiling(True)
    kacc = {}
    prof_steps = min(steps, 20)
    for _ in range(prof_steps):
        render()
        mirt.sync()
        for k, v in mirt.stats()["kernel_ms"].items():
            kacc[k] = kacc.get(k, 0.0) + v
    mirt.set_profiling(False)
    kernel_ms = {k: v / prof_steps for k, v in kacc.items() if v > 0}

    if rank == 0:
        ms_per_step = dt / steps * 1e3
        out = {
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "data": "synthetic",
            "frames_per_s": round(steps / dt, 3), "frames_in_flight": 2 if world == 1 else 1,
            "gpu_ms_per_step_rank0": round(gpu_total_ms / steps, 5),
            "kernel_ms_rank0": {k: round(v, 5) for k, v in kernel_ms.items()},
        }
        px = W * H
        if kind == "rt":
            out.update({
                "metric": "Mrays/s (primary+shadow)", "unit": "Mrays/s", "dtype": "f32",
                "value": round(rays_frame / (dt / steps) / 1e6, 3),
                "config": {"workload": args.workload, "scene": "cornell-30" if scene[0] == "cornell" else "soup-%d-seed%d" % (scene[2], scene[1]),
                           "triangles": int(len(tris)), "width": W, "height": H, "lights": 1, "soft_shadow_samples": soft_samples, "aa_samples": aa, "dof_kernel": dof,
                           "primary_rays": W * H * aa * aa, "shadow_rays": int(shadow_frame), "mode": ["auto", "brute", "binned"][st["mode_used"]],
                           "parallelism": ("bands%d+gather" % world + ("x%d" % batch if batch > 1 else "")) if world > 1 else "1gpu"},
            })
            kt = kernel_ms.get("trace", 0.0)
            kname = {mirt.RT_BRUTE: ("k_rt_tile<" if aa > 1 else "k_rt_tile2") if len(tris) <= 64 else ("k_rt_small" if len(tris) * 48 * 3 + 16 <= 48 * 1024 else "k_rt_brute"),
                     mirt.RT_BINNED: "k_rt_binned"}[st["mode_used"]]
            # algorithmic flops per launch = ray-triangle tests the launch executed x 60 flop per test as written in
            # the reference (brute force: rays x triangles; binned: candidates actually tested, counted in-kernel)
            algo_flop = tests_rank * FLOP_PER_TEST
            ach = algo_flop / (kt * 1e-3) / 1e12 if kt > 0 else None
            out["roofline"] = {"bound": "valu", "kernel": kname, "achieved": None if ach is None else round(ach, 3),
                               "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": None if ach is None else round(ach / PEAK_FP32_TFLOPS, 4),
                               "traffic": measured_traffic(args.workload, [kname]) if world == 1 else None,
                               "traffic_source": "profiles/r01_hbm_traffic.json (rocprofv3 PMC, bytes per launch)",
                               "tests_per_launch": int(tests_rank), "kernel_ms": round(kt, 5),
                               "reference_tests_per_launch": int(rays_rank * len(tris)),
                               "reference_equivalent_tflops": None if kt <= 0 else round(rays_rank * len(tris) * FLOP_PER_TEST / (kt * 1e-3) / 1e12, 3),
                               "note": "FP32 VALU-bound: not a contraction, so no MFMA; peak counts an FMA as 2 flop but bit-exact "
                                       "parity forbids FMA contraction, so the reachable ceiling is 1/2 of peak. achieved = tests the "
                                       "launch EXECUTED x 60 flop; reference_equivalent = the brute-force work of the reference "
                                       "(rays x triangles x 60) over the same time"}
            insts = measured_valu_instructions(args.workload, kname) if world == 1 else None
            if insts and kt > 0:
                # issue-slot view of the same kernel: VALU instructions per launch (profiled) over the live duration, per SIMD
                # and clock, against the measured issue ceiling of gfx950 (profiles/r01_ubench_valu_lds.txt: 0.24 / clk / SIMD)
                ipc = insts / (kt * 1e-3 * 2.4e9 * 1024)
                out["roofline"]["valu_issue"] = {"instructions_per_launch": int(insts), "achieved": round(ipc, 4), "peak": 0.24,
                                                 "unit": "wave-instr/clk/SIMD", "frac": round(ipc / 0.24, 4),
                                                 "source": "profiles/r01_pmc_issue_%s.txt (SQ_INSTS_VALU)" % args.workload}
            algo_bytes = 4.0 * W * (y1 - y0) + 60.0 * len(tris)
            if kt > 0:
                out["roofline_hbm"] = {"bound": "hbm", "achieved": round(algo_bytes / (kt * 1e-3) / 1e9, 3), "peak": PEAK_HBM_GBS,
                                       "unit": "GB/s", "frac": round(algo_bytes / (kt * 1e-3) / 1e9 / PEAK_HBM_GBS, 5),
                                       "traffic": measured_traffic(args.workload, [kname]) if world == 1 else None,
                                       "algorithmic_bytes": int(algo_bytes),
                                       "note": "algorithmic bytes = 4*W*rows framebuffer + 60*N triangle list; far below the HBM roof by construction"}
        else:
            out.update({
                "metric": "frames/s (rasteriser)", "unit": "frames/s", "dtype": "f32", "value": round(steps / dt, 3),
                "config": {"workload": args.workload, "scene": "cornell-30", "triangles": int(len(tris)), "visible_triangles": int((culled == 0).sum()),
                           "width": W, "height": H, "lights": 1, "dof_kernel": dof, "parallelism": "bands%d+gather" % world if world > 1 else "1gpu"},
            })
            # Dominant kernel: k_raster_resolve (reads the 8-byte depth key of every pixel, writes the XRGB word; it also
            # re-zeroes the keys it consumed, which replaced the per-frame clear).  Algorithmic bytes per launch = 12 B/px.
            band_px = W * (y1 - y0)
            tr = kernel_ms.get("raster_resolve", 0.0)
            if tr > 0:
                rb = 12.0 * band_px
                out["roofline"] = {"bound": "hbm", "kernel": "k_raster_resolve", "achieved": round(rb / (tr * 1e-3) / 1e9, 3),
                                   "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(rb / (tr * 1e-3) / 1e9 / PEAK_HBM_GBS, 5),
                                   "traffic": measured_traffic(args.workload, ["k_raster_resolve"]) if world == 1 else None,
                                   "traffic_source": "profiles/r01_hbm_traffic.json (rocprofv3 PMC, bytes per launch)",
                                   "algorithmic_bytes": int(rb), "kernel_ms": round(tr, 5)}
            # The whole frame: clear 8/px + fragments x 8 + resolve read 8/px + XRGB write 4/px, over the frame
            # time of the timed loop (frames overlap: the latency-bound setup of one hides behind the HBM kernels of the other)
            frag = 1.5 * px
            algo_bytes = (8 + 8 + 4) * px + 8 * frag
            out["roofline_frame"] = {"bound": "hbm", "achieved": round(algo_bytes / (ms_per_step * 1e-3) / 1e9, 3), "peak": PEAK_HBM_GBS,
                                     "unit": "GB/s", "frac": round(algo_bytes / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS, 5),
                                     "traffic": measured_traffic(args.workload, ["k_raster", "k_scan", "__amd_rocclr_fillBuffer"]) if world == 1 else None,
                                     "algorithmic_bytes": int(algo_bytes),
                                     "kernel_ms_sum": round(sum(kernel_ms.get(k, 0.0) for k in ("clear", "raster_setup", "raster_frag", "raster_resolve")), 5)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(kind, tris, culled, W, H, cam, rot, focal, samples=soft_samples, jitter=soft_jitter, aa=aa)
            if dof:
                out["cpu_baseline"]["sample"] += "; per-pixel path only, the depth-of-field blur is not part of the CPU sample"
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    mirt.shutdown()


if __name__ == "__main__":
    main()
