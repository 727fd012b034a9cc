#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native render path (see BASELINE.json / SURVEY section 8(d)).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--static-camera]

Prints ONE JSON line (rank 0).  Inputs are resident in HBM when the timed region starts.

A STEP is a batch of `frames_per_step` frames of the hot path (chosen so that the K timed steps last >= ~60 ms: a single
Cornell frame takes 25 us, and 20 of them would time launch noise).  The camera MOVES: every frame has its own yaw (an orbit
of 1 mrad per frame, as holding an arrow key does in the reference's Update(), raytracer.cpp:357-374), so whatever a frame
sizes from the view -- the binned ray tracer's pair list, the rasteriser's row tables -- is re-derived inside the timed
region, as the reference's own loop pays it (Draw() runs only when the view changed, :165-173).  `static_camera` reports the
same loop with one fixed view beside it.

Workloads (BASELINE.json `configs`):
    soup100k     (default; configs[2], the north-star target) 100k random triangles, 1920x1080, one light
    cornell1080  (configs[1], the config the reference's own number would be quoted on) Cornell box, 1920x1080, primary + shadow rays
    raster4k     (configs[3]) rasteriser, Cornell box, 3840x2160
    cornell500   (configs[0]) the reference's own 500x500 case
    soup1m8k     (configs[4]) 1M random triangles, 7680x4320 (meant for 8 GPUs)
    cornell1080soft16 / cornell1080aa3 / cornell1080dof8 / raster4kdof8   SURVEY 8(f) ranks 1-3 switched on
With no --workload the line is soup100k (binned: bit-identical to brute force, tests/test_gpu_baseline_configs.py) and carries
`sub_results` for cornell1080, raster4k, soup1m8k and `soup100k_brute` -- ONE frame of the same workload through the LDS-tiled
brute-force kernel, the literal wording of configs[2] and the one kernel whose flop count is the reference's (rays x triangles x 60).

With --gpus N > 1 (`python bench.py --gpus N` starts its own ranks; under the driver's torch.distributed.run it reads RANK /
LOCAL_RANK / WORLD_SIZE) the frame is split into N bands of rows -- of equal estimated cost where the frame is binned
(MIRT_PARTITION_WEIGHTED), of equal height otherwise; every rank renders its band and the XRGB bands are gathered on rank 0 over
RCCL ("scaling": "strong"); frames that render in microseconds travel 32 to a gather.  MIRT_BENCH_REHEARSAL=1 runs that control flow with every rank on device 0 (gloo,
host-staged gathers) and checks the assembled frames against a single-GPU frame -- not a measurement.
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

RASTER_FLOP_PER_PIXEL = 109.0   # interpolation of the winning fragment + PixelShader with one light, as written (rasteriser.cpp:549-589, 661-662)
FLOP_PER_TEST = 60.0          # 57 add/mul + 3 div as written in raytracer.cpp:216-239
PEAK_FP32_TFLOPS = 157.3      # MI355X_MICROARCH.md: vector FP32 peak (counts FMA as 2; this path may not fuse)
# what bit-exact parity leaves reachable: no FMA contraction, i.e. half the peak: 78.6 -- with packed v_pk_* (two flops per lane every
# four cycles) or with one-lane v_mul / v_add on vector registers (one flop per lane every two cycles: tools/ubench.hip measures 0.43 of
# the nominal 0.5 such instructions per clock and SIMD, and 0.232 packed ones of 0.25).  A one-lane instruction with a scalar-register
# operand issues every four cycles (39.3): what "no_fma_scalar" stood for through round 3, when only that case had been probed.
PEAK_REACHABLE_TFLOPS = {"no_fma_packed": 78.6, "no_fma_one_lane_vector_operands": 78.6, "no_fma_one_lane_scalar_operand": 39.3}
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec peak
ISSUE_CEILING = 0.24          # issue ceiling of FOUR-CYCLE vector instructions, wave-instr / clk / SIMD (profiles/r04_ubench.txt): the fallback
ROUND = "r04"                 # which committed profiles/ files `traffic` and `valu_issue` are read from
ORBIT_STEP = 1.0e-3           # yaw per frame of the moving camera (rad)
ORBIT_VIEWS = 64              # distinct views cycled through (consecutive frames never share one)

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
DEFAULT_WORKLOAD = "soup100k"                              # the north star's target configuration (BASELINE configs[2]: 1080p, 100 k triangles)
SUB_RESULTS = ("cornell1080", "raster4k", "soup1m8k")     # embedded in the default line
SUB_RESULTS_SHARDED = ("soup1m8k",)                       # ... of a --gpus N > 1 run: BASELINE configs[4], the config the band split is for


HOST_ONLY_SOURCES = ("mirt_capi.hip", "comm.cpp", "comm.hpp", "scene_host.cpp")      # no device code a counter could have counted


def csrc_digest():
    """sha256 (first 16 hex digits) over the DEVICE code's text -- the kernel sources and their headers, comments and blank space
    stripped, so that rewording a comment or touching host code does not disown the counters -- : what a committed PMC summary must
    have been profiled at.  Taken ON THE GPU BOX when the counters are collected (tools/collect_profiles.sh writes it beside each
    summary; tools/store_profiles.py only copies it), never recomputed when the files are stored."""
    import hashlib
    import re
    h = hashlib.sha256()
    d = os.path.join(ROOT, "cpp-raytracer-rasterizer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".cpp")) and name not in HOST_ONLY_SOURCES:
            h.update(name.encode())
            with open(os.path.join(d, name), "r", errors="replace") as f:
                text = f.read()
            text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)        # block comments
            text = re.sub(r"//[^\n]*", "", text)                       # line comments (no string literal in csrc/ holds "//")
            h.update(" ".join(text.split()).encode())
    return h.hexdigest()[:16]


_stale_warned = set()


def committed_profile(name, workload):
    """The counters of one workload from a committed PMC summary -- or {} with a warning, when they were collected at other
    kernel sources than the ones running (`stamps[workload]`, the digest taken on the GPU box at collection time): a counter read
    from a file must not outlive the kernel it counted."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        return {}
    if workload not in doc.get("workloads", {}):
        return {}
    stamp = doc.get("stamps", {}).get(workload, doc.get("csrc_sha16"))
    if stamp != csrc_digest():
        if (name, workload) not in _stale_warned:
            _stale_warned.add((name, workload))
            print("bench.py: profiles/%s [%s] was collected at kernel sources %s, running %s: its counters are reported as null "
                  "(re-run tools/collect_profiles.sh)" % (name, workload, stamp, csrc_digest()), file=sys.stderr, flush=True)
        return {}
    return doc["workloads"][workload]


def measured_traffic(workload, kernel_prefixes, per_frame=False):
    """HBM bytes per launch of the named kernel(s) from the committed rocprofv3 PMC summary (profiles/<round>_hbm_traffic.json:
    separate FETCH_SIZE / WRITE_SIZE passes, gfx950 FETCH correction applied).  bench.py cannot run the profiler on itself,
    so `traffic` is the last profiled value for this exact workload, or None.  per_frame: the sum over the kernels of a FRAME --
    those the pass saw more than twice (the per-scene table builders, the one-off light-cube build and torch's fills run once or
    twice), each weighted by its launches per frame."""
    w = committed_profile("%s_hbm_traffic.json" % ROUND, workload)
    ks = one_instantiation({k: v for k, v in w.items() if any(k.replace("mirt::", "").replace("void ", "").startswith(p) for p in kernel_prefixes)})
    if per_frame:
        ks = {k: v for k, v in ks.items() if v.get("launches", 0) > 2}
        most = max((v["launches"] for v in ks.values()), default=0)
        # (weights as in measured_valu_per_frame: nearly as often as the kernel launched most = once per frame of the loop)
        tot = sum((v["fetch_bytes"] + v["write_bytes"]) * (1.0 if v["launches"] >= 0.75 * most else v["launches"] / most) for v in ks.values()) if most else 0
    else:
        tot = sum(v["fetch_bytes"] + v["write_bytes"] for v in ks.values())
    return int(tot) or None


def one_instantiation(kernels):
    """Of the instantiations of one kernel template a pass saw (k_rt_trace2<false, false> in the frames of the loop, <false, true> --
    the one that keeps statistics -- in the profiled frames) only the one launched most: they are the same launch counted twice."""
    best, total = {}, {}
    for k, v in kernels.items():
        base = k.split("<")[0]
        total[base] = total.get(base, 0) + v.get("launches", 0)
        if base not in best or v.get("launches", 0) > best[base][1].get("launches", 0):
            best[base] = (k, v)
    # (the counters of the instantiation launched most, the launches of all of them: how often the KERNEL ran per frame)
    return {k: dict(v, launches=total[k.split("<")[0]]) if "launches" in v else v for k, v in best.values()}


def measured_valu_instructions(workload, kernel_prefix):
    """Wave-level VALU instructions per launch of the named kernel from the committed rocprofv3 PMC pass
    (profiles/<round>_pmc_issue.json: SQ_INSTS_VALU averaged over the launches), or None."""
    for k, v in one_instantiation(committed_profile("%s_pmc_issue.json" % ROUND, workload)).items():
        if kernel_prefix in k and "SQ_INSTS_VALU" in v:
            return float(v["SQ_INSTS_VALU"])
    return None


def measured_counter(workload, kernel_prefix, counter):
    """Any counter of the named kernel from the same committed pass (per launch), or None."""
    for k, v in one_instantiation(committed_profile("%s_pmc_issue.json" % ROUND, workload)).items():
        if kernel_prefix in k and counter in v:
            return float(v[counter])
    return None


_issue_mix = None


def issue_ceiling(kernel_prefix):
    """The vector issue ceiling of one kernel, wave-instr / clk / SIMD: its instructions weighted by what each CLASS costs on this chip
    -- v_mul / v_add / v_fma / v_mov on vector registers 2.35 cycles, integer add / logic 2.9, everything else (packed, compares,
    selects, conversions, shifts, anything reading a scalar register) 4.24, reciprocal / square root 8.3: tools/ubench.hip,
    profiles/r04_ubench.txt -- over the kernel's TEXT (tools/issue_mix.py -> profiles/<round>_issue_mix.json: a static mix, loops not
    weighted).  Rounds 1-3 priced every instruction at four cycles (0.24), which the fast class beats; ISSUE_CEILING remains the
    fallback when the table is missing or was made from other kernel sources."""
    global _issue_mix
    if _issue_mix is None:
        _issue_mix = {}
        path = os.path.join(ROOT, "profiles", "%s_issue_mix.json" % ROUND)
        try:
            with open(path) as f:
                doc = json.load(f)
            if doc.get("csrc_sha16") == csrc_digest():
                _issue_mix = doc.get("kernels", {})
            else:
                sys.stderr.write("bench.py: %s was made from other kernel sources; issue ceilings fall back to %.2f\n" % (path, ISSUE_CEILING))
        except (OSError, ValueError):
            pass
    names = sorted(k for k in _issue_mix if kernel_prefix in k)
    # (the instantiation the benchmark lines run: no supersampling, no statistics)
    names.sort(key=lambda k: (k != "mirt::" + kernel_prefix, "<false, false" not in k, "true" in k))
    return float(_issue_mix[names[0]]["ceiling"]) if names else ISSUE_CEILING


def valu_plus_salu_issue(workload, kernel_prefix, kernel_ms):
    """Vector AND scalar instructions per launch (SQ_INSTS_VALU + SQ_INSTS_SALU) over the kernel's duration, per SIMD and clock, against
    the measured issue ceiling of a vector stream: at the four or five waves per SIMD these kernels run at a scalar instruction is not
    free (profiles/r03_issue_model.txt), so the two together are what the issue side is offered."""
    nv, ns = measured_counter(workload, kernel_prefix, "SQ_INSTS_VALU"), measured_counter(workload, kernel_prefix, "SQ_INSTS_SALU")
    if not nv or ns is None or kernel_ms <= 0:
        return None
    ipc = (nv + ns) / (kernel_ms * 1e-3 * 2.4e9 * 1024)
    peak = issue_ceiling(kernel_prefix)
    return {"valu_per_launch": int(nv), "salu_per_launch": int(ns), "achieved": round(ipc, 4), "peak": peak, "frac": round(ipc / peak, 4),
            "unit": "vector + scalar wave-instr/clk/SIMD", "source": "profiles/%s_pmc_issue.json (SQ_INSTS_VALU + SQ_INSTS_SALU)" % ROUND}


def all_kinds_issue(workload, kernel_prefix, kernel_ms):
    """Issue-active slots of ANY instruction kind (SQ_ACTIVE_INST_ANY: vector, scalar, LDS, memory, branch, wait; equals the
    instruction count for vector instructions, counts an LDS / memory instruction that holds its pipe longer more than once) per
    launch over the kernel's duration, per SIMD and clock: how busy the issue side as a whole is."""
    n = measured_counter(workload, kernel_prefix, "SQ_ACTIVE_INST_ANY")
    if not n or kernel_ms <= 0:
        return None
    ipc = n / (kernel_ms * 1e-3 * 2.4e9 * 1024)
    peak = issue_ceiling(kernel_prefix)
    return {"instructions_per_launch": int(n), "achieved": round(ipc, 4), "vector_stream_rate": peak, "ratio": round(ipc / peak, 4),
            "unit": "issue-active slots of any instruction kind/clk/SIMD", "source": "profiles/%s_pmc_issue.json (SQ_ACTIVE_INST_ANY)" % ROUND,
            "note": "against the rate measured for pure vector streams; kinds overlap and long-held LDS / memory instructions count more than once, so a mixed stream may exceed it"}


def measured_valu_per_frame(workload):
    """VALU instructions of ALL per-frame kernels of the workload (every kernel the PMC pass saw more than twice: the
    per-scene table builders run once), per frame, from the same committed pass; or None."""
    ks = one_instantiation(committed_profile("%s_pmc_issue.json" % ROUND, workload))
    per_frame = [v for v in ks.values() if "SQ_INSTS_VALU" in v and v.get("launches", 0) > 2]
    if not per_frame:
        return None
    most = max(v["launches"] for v in per_frame)
    # (a kernel launched nearly as often as the one launched most runs once per frame of the loop: the pass's warm-up frames, whose view
    # stands still, skip the binning kernels; one launched rarely -- a light-cube build -- counts by its share)
    return sum(float(v["SQ_INSTS_VALU"]) * (1.0 if v["launches"] >= 0.75 * most else v["launches"] / most) for v in per_frame)


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
    cores = min(cores, 64)                      # OpenMP over pixels stops scaling long before 256 threads here
    centre = H // 2
    t0 = time.perf_counter()
    soft = dict(samples=samples, jitter=jitter, aa=aa)
    r = o.raytrace(tris, cam, rot, focal, W, H, LIGHT, y0=centre, y1=centre + 1, threads=cores, want=("xrgb",), **soft)
    probe = max(time.perf_counter() - t0, 1e-4)
    rows = int(max(1, min(H, budget_s / probe)))
    if rows >= H:
        # the whole frame fits the budget: repeat it until ~10-30 s of CPU work (threads x wall) have gone by -- one 13 ms frame
        # measures thread start-up more than rays -- and report the mean rate
        reps, t0 = 0, time.perf_counter()
        while True:
            r = o.raytrace(tris, cam, rot, focal, W, H, LIGHT, threads=cores, want=("xrgb",), **soft)
            reps += 1
            dt = time.perf_counter() - t0
            if dt * cores >= 20.0 or dt >= min(budget_s, 3.0) or reps >= 200:
                break
        rays = (W * H * aa * aa + r["nshadow"]) * reps
        sample = "%d full %dx%d frame%s" % (reps, W, H, "s" if reps > 1 else "")
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


class Env:
    """Process-wide state: ranks, device, streams."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        import mirt
        self.torch, self.dist, self.mirt = torch, dist, mirt
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        args.gpus = self.world                      # (main() has started the ranks itself when --gpus asked for more than the environment holds)
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the render path has no CPU fallback")
        # MIRT_BENCH_REHEARSAL=1: every rank on device 0 with a gloo group and host-staged gathers -- the N > 1 control
        # flow (batches, events, reductions) on a one-GPU box.  Not a measurement.
        self.rehearsal = self.world > 1 and os.environ.get("MIRT_BENCH_REHEARSAL") == "1"
        if self.rehearsal:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.rehearsal:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", local_rank))
        if self.rehearsal:
            os.environ["MIRT_COMM"] = "shm"                 # ranks share device 0: the library's host-staged loopback transport
        mirt.init(local_rank)
        # The gather of the bands is the library's own (mirt_*_sharded: RCCL point-to-point inside libmirt.so, loaded at run
        # time); torch.distributed only carries the 128-byte group id to the ranks.  MIRT_BENCH_GATHER=torch keeps the gather
        # in Python (mirt/sharding.py, torch.distributed.gather) instead.
        self.native = self.world > 1 and os.environ.get("MIRT_BENCH_GATHER", "native") != "torch"
        self.native_error = None
        if self.native:
            # every rank reports whether its side of the group came up (id, communicator, a send + receive to itself through the
            # transport); if any did not, all ranks fall back to the Python gather together rather than lose the run
            try:
                ids = [mirt.comm_create_id() if self.rank == 0 else None]
            except mirt.MirtError as e:
                ids, self.native_error = [None], str(e)
            dist.broadcast_object_list(ids, src=0)
            if ids[0] is not None and self.native_error is None:
                try:
                    mirt.comm_init(ids[0], self.rank, self.world)
                    mirt.comm_selfcheck(1 << 20)
                except mirt.MirtError as e:
                    self.native_error = str(e)
            elif self.native_error is None:
                self.native_error = "rank 0 could not create the group id"
            flag = torch.tensor([0.0 if self.native_error is None else 1.0], device="cpu" if self.rehearsal else torch.device("cuda", local_rank))
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if flag.item() != 0.0:
                if self.native_error is None:
                    self.native_error = "another rank could not join the group"
                print("bench.py rank %d: library gather unavailable (%s); falling back to torch.distributed.gather" % (self.rank, self.native_error), file=sys.stderr, flush=True)
                try:
                    mirt.comm_shutdown()
                except mirt.MirtError:
                    pass
                self.native = False
        self.dev = torch.device("cuda", local_rank)
        self.mirt_stream = torch.cuda.ExternalStream(mirt.load().mirt_stream(), device=self.dev)
        self.comm_stream = torch.cuda.Stream(device=self.dev) if self.world > 1 else None

    def fence(self):
        self.mirt.sync()
        if self.comm_stream is not None:
            self.comm_stream.synchronize()
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def reduce(self, values, op):
        if self.world == 1:
            return list(values)
        rdev = self.torch.device("cpu") if self.rehearsal else self.dev
        t = self.torch.tensor(list(values), dtype=self.torch.float64, device=rdev)
        self.dist.all_reduce(t, op=op)
        return [float(x) for x in t.tolist()]


def run_workload(env, name, steps, warmup, mode_name, moving, want_cpu, target_s=0.06, extras=True, quick=False):
    """Times `steps` steps of workload `name` and returns its result dict (rank 0; None elsewhere).  quick: a handful of frames
    around the timed ones instead of dozens (frames that take a sixth of a second: the brute-force kernel on 100 k triangles)."""
    torch, dist, mirt = env.torch, env.dist, env.mirt
    from mirt.sharding import BandGather
    world, rank, dev = env.world, env.rank, env.dev
    kind, scene, W, H, cam, focal, rot11 = WORKLOADS[name]
    tris = mirt.scene_cornell() if scene[0] == "cornell" else mirt.scene_soup(scene[1], scene[2], scene[3])
    nviews = ORBIT_VIEWS if moving else 1
    views = [mirt.make_view(cam, mirt.rot_from_yaw(i * ORBIT_STEP, rot11), focal, W, H) for i in range(nviews)]
    mirt.set_frames_in_flight(1)
    mirt.scene_upload(tris, None)
    culled0 = mirt.cull(tris, views[0], 3) if kind == "raster" else None
    if kind == "raster":
        mirt.scene_set_culled(culled0)
    mode = {"auto": mirt.RT_AUTO, "brute": mirt.RT_BRUTE, "binned": mirt.RT_BINNED}[mode_name]
    soft_samples, soft_jitter = 1, None
    if name.endswith("soft16"):
        # AddLight's jitter (raytracer.cpp:186-190): 16 positions, each coordinate light + U[-0.04, 0.04] (synthetic here:
        # a fixed-seed numpy stream instead of the C library's rand())
        soft_samples = 16
        soft_jitter = (LIGHT[:, 0:3] + (np.random.RandomState(1).rand(16, 3).astype(np.float32) - np.float32(0.5)) * np.float32(0.08)).astype(np.float32)
    mirt.set_soft_shadows(soft_samples, soft_jitter)
    aa = 3 if name.endswith("aa3") else 1
    mirt.set_antialiasing(aa)
    dof = 8 if name.endswith("dof8") else 0
    mirt.set_depth_of_field(dof, 1.3 if kind == "rt" else 1.9)

    # One GPU: several frames in flight, each into its own band buffer (the library takes its streams in turn, so the next
    # frames are dispatched -- and fill the device's gaps -- while the previous ones drain).  Several GPUs: one frame in
    # flight per rank, the RCCL gather of the previous batch overlaps it instead.
    in_flight = int(os.environ.get("MIRT_BENCH_IN_FLIGHT", "4")) if world == 1 else 1
    if quick:
        in_flight = 1                                   # (frames of a sixth of a second each fill the chip alone)
    depth = max(2, in_flight)
    # Several GPUs: frames that render faster than a collective starts (the 30-triangle scenes) travel `batch` at a time --
    # one RCCL gather moves the bands of 32 consecutive frames; heavy frames (the soups: milliseconds) go one per gather.
    # Every frame is still rendered, gathered and assembled inside the timed region.
    batch = 32 if (world > 1 and len(tris) < 1000 and kind == "rt") else 1
    native = getattr(env, "native", False)
    # the library's own split: bands of equal estimated cost where the frame is binned (every rank derives them from the cost
    # histogram of an earlier frame, nothing exchanged), equal rows otherwise
    weighted = native and kind == "rt" and len(tris) >= 65 and os.environ.get("MIRT_BENCH_PARTITION", "weighted") == "weighted"
    if world > 1:
        mirt.set_partition(mirt.PARTITION_WEIGHTED if weighted else 0)
    bands = BandGather(H, W, dev, depth=depth, batch=1 if native else batch, via_host=env.rehearsal)
    y0, y1 = bands.y0, bands.y1
    mirt.set_frames_in_flight(in_flight)
    cull_per_frame = kind == "raster" and moving        # the rasteriser's Update() culls for the new view (rasteriser.cpp:404-447)
    renders = {}                                          # (view, buffer, slot) -> callable that enqueues the frame

    def render_fn(v, k, b):
        key = (v, k, b)
        if key not in renders:
            ptr = bands.slot(k, b).data_ptr()
            if kind == "rt":
                renders[key] = mirt.prepared_raytrace_device(views[v], LIGHT, INDIRECT, mode, y0, y1, y0, ptr, W * 4)
            else:
                renders[key] = mirt.prepared_rasterise_device(views[v], LIGHT, INDIRECT, y0, y1, y0, ptr, W * 4)
        return renders[key]

    cullers = [mirt.prepared_cull_device(v, 3) for v in views] if cull_per_frame else None
    mirt_stream, comm_stream = env.mirt_stream, env.comm_stream
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

    # native sharding: one library call renders this rank's band of `batch` consecutive views and gathers them on rank 0
    root_frames = [torch.zeros((batch, H, W), dtype=torch.int32, device=dev) for _ in range(depth)] if (native and rank == 0) else None
    sharded = {}

    def sharded_fn(v0, k):
        key = (v0, k)
        if key not in sharded:
            vs = [views[(v0 + b) % nviews] for b in range(batch)]
            sharded[key] = mirt.prepared_sharded(kind, vs, LIGHT, INDIRECT, mode, 0, root_frames[k].data_ptr() if root_frames else None, W * 4)
        return sharded[key]

    def frame():
        i = frame_no[0]
        frame_no[0] += 1
        v = i % nviews
        if native:
            if i % batch == 0:
                if cullers is not None:
                    cullers[v]()
                sharded_fn(v, (i // batch) % depth)()
            return
        if cullers is not None:
            cullers[v]()
        if world == 1:
            render_fn(v, i % depth, 0)()
            return
        # Double-buffered band buffers: batch j renders into buffer j%2 on mirt's stream while the RCCL gather of batch
        # j-1 (the other buffer) is still in flight on the communication stream.  Dependencies are two events per buffer.
        b, k = i % batch, (i // batch) % depth
        if b == 0:
            mirt_stream.wait_event(gathered[k])                # the gather that last read buffer k has finished
        render_fn(v, k, b)()
        if b == batch - 1:
            flush(k)

    last_batch = [batch]                             # frames the most recent gather carried

    def finish_batch():
        """Gathers a batch the loop left incomplete and restarts the batch numbering."""
        last_batch[0] = batch
        if native:                                       # (frames are issued in whole batches)
            frame_no[0] = 0
            return
        if world > 1 and frame_no[0] % batch:
            last_batch[0] = frame_no[0] % batch
            flush((frame_no[0] // batch) % depth)
        frame_no[0] = 0

    def timed(nframes):
        env.fence()
        t0 = time.perf_counter()
        for _ in range(nframes):
            frame()
        finish_batch()
        env.fence()
        return time.perf_counter() - t0

    # calibrate the batch: frames per step so that the timed region lasts >= target_s (agreed over the ranks)
    timed(2 if quick else max(2 * batch, 12))          # (also past the frames after which the light-cube grid settles: mirt_capi light_cache_ensure)
    probe_frames = 2 if quick else max(2 * batch, 8)
    t_frame = timed(probe_frames) / probe_frames
    t_frame = env.reduce([t_frame], dist.ReduceOp.MAX)[0] if world > 1 else t_frame
    fps_step = max(1, int(np.ceil(1.3 * target_s / (steps * max(t_frame, 1e-7)))))       # (a third over: the probe's frames run a little slower than the loop's)
    if world > 1:
        fps_step = ((fps_step + batch - 1) // batch) * batch
    for w in range(warmup):
        dw = timed(fps_step)
        if w == 0 and target_s > 0:
            # the probe's few frames overestimate what a long loop sustains for the fastest frames (the Cornell box at 500 x 500 takes 5 us):
            # the first warm-up step says what a step really takes, and the steps grow until the timed region will last target_s
            dw = env.reduce([dw], dist.ReduceOp.MIN)[0] if world > 1 else dw
            if dw * steps < target_s:
                fps_step = int(np.ceil(fps_step * 1.2 * target_s / max(dw * steps, 1e-9)))
                if world > 1:
                    fps_step = ((fps_step + batch - 1) // batch) * batch
    dt = timed(steps * fps_step)
    st = mirt.stats()
    dt = env.reduce([dt], dist.ReduceOp.MAX)[0] if world > 1 else dt
    rays_rank = float(st["primary_rays"] + st["shadow_rays"]) if kind == "rt" else 0.0
    rays_frame, tests_frame, shadow_frame = env.reduce([rays_rank, float(st["tests"]), float(st["shadow_rays"])], dist.ReduceOp.SUM)

    if env.rehearsal and rank == 0:
        # the frames rank 0 assembled from every rank's bands against one full-frame render of the same view (interior
        # words; the border is never written by the ray tracer)
        vlast = (steps * fps_step - 1) % nviews
        full = torch.zeros((H, W), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()                         # (torch fills on ITS stream; the library's streams are not ordered with it)
        if cullers is not None:
            cullers[vlast]()
        if kind == "rt":
            mirt.raytrace_device(views[vlast], LIGHT, INDIRECT, mode, 0, H, 0, full.data_ptr(), W * 4)
        else:
            mirt.rasterise_device(views[vlast], LIGHT, INDIRECT, 0, H, 0, full.data_ptr(), W * 4)
        mirt.sync()
        if native:
            got = root_frames[((steps * fps_step - 1) // batch) % depth][(steps * fps_step - 1) % batch].clone()
            if kind == "rt":                             # border words: never written by the ray tracer (received bands carry 0)
                got[0, :] = full[0, :]; got[-1, :] = full[-1, :]; got[:, 0] = full[:, 0]; got[:, -1] = full[:, -1]
        else:
            got = bands.frames[last_batch[0] - 1] if batch > 1 else bands.frame
        same = bool(torch.equal(got, full))
        print("rehearsal %s: the last gathered frame is identical to the single-GPU frame: %s" % (name, same), file=sys.stderr, flush=True)
        if not same:
            bad_rows = torch.nonzero((got != full).any(dim=1)).flatten()
            raise SystemExit("rehearsal: gathered frame differs from the single-GPU frame (%d words in %d rows, first row %d, last row %d; bands %s)"
                             % (int((got != full).sum()), int(bad_rows.numel()), int(bad_rows[0]), int(bad_rows[-1]), [mirt.band_of(r, world, H) for r in range(world)]))

    # per-kernel durations of this rank IN THE SAME MODE as the timed loop (same frames in flight, same moving camera):
    # the library brackets every launch with hipEvents on the stream it runs on
    # -- with two frames in flight the frame measured is the one BEFORE the last of a run of frames queued back to back
    # (mirt_get_previous_kernel_ms): its kernels ran between the frames on both sides of it, as every frame of the timed loop
    # does; the last frame of a run drains alone and would read like a single-stream launch.
    mirt.set_profiling(True)
    kacc, kn = {}, 0
    overlapped = in_flight >= 2 and not native and batch == 1
    for i in range(1 if quick else max(4, min(steps * fps_step, 64) // 4)):
        for _ in range((3 if quick else 6) if overlapped else 1):
            frame()
        if overlapped:
            kms, _ = mirt.previous_kernel_ms()
        else:
            mirt.sync()
            kms = mirt.stats()["kernel_ms"]
        for k, v in kms.items():
            kacc[k] = kacc.get(k, 0.0) + v
        kn += 1
    finish_batch()
    env.fence()
    kernel_ms = {k: v / kn for k, v in kacc.items() if v > 0}
    # ... and the same launches ALONE on the device (a sync after every frame): what the kernel takes when nothing shares the
    # chip with it -- beside the line's figures, never in them
    kernel_ms_alone = None
    if overlapped:
        aacc, an = {}, 0
        for i in range(2 if quick else 12):
            frame()
            mirt.sync()
            for k, v in mirt.stats()["kernel_ms"].items():
                aacc[k] = aacc.get(k, 0.0) + v
            an += 1
        finish_batch()
        env.fence()
        kernel_ms_alone = {k: v / an for k, v in aacc.items() if v > 0}
    # the binned kernel keeps its own counts (tests executed, candidates offered) only for frames rendered with profiling on --
    # five scalar instructions per filter step that the frames of the timed loop do without: the last profiled frame has them
    mirt.sync()
    st_prof = mirt.stats()
    mirt.set_profiling(False)

    static = None
    if extras and moving and world == 1:
        # the same loop with ONE fixed view: the sizing caches hit, nothing is read back per frame
        v0 = views[0]
        fns = [mirt.prepared_raytrace_device(v0, LIGHT, INDIRECT, mode, y0, y1, y0, bands.slot(k, 0).data_ptr(), W * 4) if kind == "rt"
               else mirt.prepared_rasterise_device(v0, LIGHT, INDIRECT, y0, y1, y0, bands.slot(k, 0).data_ptr(), W * 4) for k in range(2)]
        if kind == "raster":
            mirt.cull_device(v0, 3)
        n = max(8, min(steps * fps_step, int(0.05 / max(t_frame, 1e-7))))
        for i in range(8):
            fns[i & 1]()
        env.fence()
        t0 = time.perf_counter()
        for i in range(n):
            fns[i & 1]()
        env.fence()
        sdt = (time.perf_counter() - t0) / n
        static = {"ms_per_frame": round(sdt * 1e3, 5), "frames_per_s": round(1.0 / sdt, 2), "frames": n}
        if kind == "rt":
            static["value"] = round(rays_frame / sdt / 1e6, 3)

    host = None
    if extras and world == 1:
        # frames/s INCLUDING the framebuffer's way to the host -- the entry point Draw() binds
        # (mirt_raytrace / mirt_rasterise into the caller's host surface), never `value`
        host = host_path_rate(mirt, kind, views, mode, W, H, t_frame)

    covered = None
    if kind == "raster" and rank == 0 and world == 1:
        z = torch.zeros((H, W), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()                             # (torch fills on ITS stream; the library's streams are not ordered with it)
        mirt.cull_device(views[0], 3)
        mirt.rasterise_device(views[0], LIGHT, INDIRECT, 0, H, 0, bands.slot(0, 0).data_ptr(), W * 4, None, z.data_ptr(), None)
        mirt.sync()
        covered = int((z > 0).sum().item())

    out = None
    if rank == 0:
        nframes = steps * fps_step
        ms_frame = dt / nframes * 1e3
        out = {
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 5),
            "frames_per_step": fps_step, "ms_per_frame": round(ms_frame, 5), "timed_region_s": round(dt, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "data": "synthetic",
            "frames_per_s": round(nframes / dt, 3), "frames_in_flight": in_flight,
            "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)"),
            "camera": ("orbit: yaw += %g rad per frame, %d views" % (ORBIT_STEP, nviews)) if moving else "static",
            "kernel_ms_rank0": {k: round(v, 5) for k, v in kernel_ms.items()},
        }
        if kernel_ms_alone:
            out["kernel_ms_alone_rank0"] = {k: round(v, 5) for k, v in kernel_ms_alone.items()}
        if static:
            out["static_camera"] = static
        if host:
            out["host_path"] = host
            # frames/s INCLUDING the framebuffer's way to the host surface; synchronous Draw() and the
            # present-one-draw-the-next loop (mirt_*_async)
            out["frames_per_s_host"] = host.get("frames_per_s")
            if "frames_per_s" in host.get("async", {}):
                out["frames_per_s_host_async"] = host["async"]["frames_per_s"]
        px = W * H
        if kind == "rt":
            out.update({
                "metric": "Mrays/s (primary+shadow)", "unit": "Mrays/s", "dtype": "f32",
                "value": round(rays_frame / (dt / nframes) / 1e6, 3),
                "config": {"workload": name, "scene": "cornell-30" if scene[0] == "cornell" else "soup-%d-seed%d" % (scene[2], scene[1]),
                           "triangles": int(len(tris)), "width": W, "height": H, "lights": 1, "soft_shadow_samples": soft_samples, "aa_samples": aa, "dof_kernel": dof,
                           "primary_rays": W * H * aa * aa, "shadow_rays": int(shadow_frame), "mode": ["auto", "brute", "binned"][st["mode_used"]],
                           "parallelism": ("bands%d%s+%s" % (world, "(cost-weighted)" if weighted else "", "rccl-p2p-gather(libmirt)" if native else "torch-gather") + ("x%d" % batch if batch > 1 else "")) if world > 1 else "1gpu"},
            })
            kt = kernel_ms.get("trace", 0.0)
            kname = {mirt.RT_BRUTE: "k_rt_tile2" if len(tris) <= 64 else ("k_rt_small" if len(tris) * 48 * 3 + 16 <= 48 * 1024 else "k_rt_brute"),
                     mirt.RT_BINNED: "k_rt_trace2"}[st["mode_used"]]
            # algorithmic flops per launch = ray-triangle tests the launch executed x 60 flop per test as written in the
            # reference (brute force: rays x triangles; tile / binned kernels: filter evaluations counted in-kernel)
            tests_rank = float(st_prof["tests"])
            ach = tests_rank * FLOP_PER_TEST / (kt * 1e-3) / 1e12 if kt > 0 else None
            out["roofline"] = {"bound": "valu", "kernel": kname, "achieved": None if ach is None else round(ach, 3),
                               "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": None if ach is None else round(ach / PEAK_FP32_TFLOPS, 4),
                               "peak_reachable": PEAK_REACHABLE_TFLOPS,
                               "frac_of_reachable": None if ach is None else round(ach / PEAK_REACHABLE_TFLOPS["no_fma_packed"], 4),
                               "primary_fraction": "valu_issue (the kernel skips most of the reference's tests, so executed-test flops "
                                                   "understate what the vector pipes do; issue slots do not)",
                               "traffic": measured_traffic(name, [kname]) if world == 1 else None,
                               "traffic_source": "profiles/%s_hbm_traffic.json (rocprofv3 PMC, bytes per launch)" % ROUND,
                               "tests_per_launch": int(tests_rank), "candidates_per_launch": int(st_prof["candidates"]), "kernel_ms": round(kt, 5),
                               "kernel_ms_mode": "hipEvents around the launch on its own stream, %d frame(s) in flight, %s camera%s" % (in_flight, "moving" if moving else "static", "; the frame before the last of 6 queued back to back (overlapped on both sides)" if overlapped else ""),
                               "reference_tests_per_launch": int(rays_rank * len(tris)),
                               "reference_equivalent_tflops": None if kt <= 0 else round(rays_rank * len(tris) * FLOP_PER_TEST / (kt * 1e-3) / 1e12, 3),
                               "note": "FP32 VALU-bound: not a contraction, so no MFMA; peak counts an FMA as 2 flop but bit-exact "
                                       "parity forbids FMA contraction (peak_reachable). achieved = filter tests the launch EXECUTED x 60 "
                                       "flop (candidates skipped by binning, depth order, the near bound, or settled as certain occluders "
                                       "do not count: the better the skipping, the lower this fraction); reference_equivalent = the "
                                       "brute-force work of the reference (rays x triangles x 60) over the same time"}
            # launches of consecutive frames overlap, so a launch's own duration understates what the chip does: the same work over
            # the FRAME time, and the launch alone on the device, beside it
            fach = tests_rank * FLOP_PER_TEST / (ms_frame * 1e-3) / 1e12
            out["roofline"]["frame"] = {"achieved": round(fach, 3), "frac": round(fach / PEAK_FP32_TFLOPS, 4), "ms_per_frame": round(ms_frame, 5)}
            if kernel_ms_alone and kernel_ms_alone.get("trace", 0.0) > 0:
                ka = kernel_ms_alone["trace"]
                out["roofline"]["alone"] = {"kernel_ms": round(ka, 5), "achieved": round(tests_rank * FLOP_PER_TEST / (ka * 1e-3) / 1e12, 3),
                                            "frac": round(tests_rank * FLOP_PER_TEST / (ka * 1e-3) / 1e12 / PEAK_FP32_TFLOPS, 4)}
            insts = measured_valu_instructions(name, kname) if world == 1 else None
            if insts and kt > 0:
                # issue-slot view of the same kernel: VALU instructions per launch (profiled) over the live duration, per SIMD
                # and clock, against the measured issue ceiling of gfx950
                ipc = insts / (kt * 1e-3 * 2.4e9 * 1024)
                peak = issue_ceiling(kname)
                out["roofline"]["valu_issue"] = {"instructions_per_launch": int(insts), "achieved": round(ipc, 4), "peak": peak,
                                                 "peak_is": "this kernel's instruction classes at their measured issue cost (tools/issue_mix.py; 0.24 = every instruction at four cycles, the figure of rounds 1-3)",
                                                 "unit": "wave-instr/clk/SIMD", "frac": round(ipc / peak, 4), "frac_at_four_cycles": round(ipc / ISSUE_CEILING, 4),
                                                 "lane_slots_per_test": round(insts * 64.0 / max(tests_rank, 1.0), 1),
                                                 "lane_slots_per_candidate": round(insts * 64.0 / max(float(st_prof["candidates"]), 1.0), 1),
                                                 "source": "profiles/%s_pmc_issue.json (SQ_INSTS_VALU)" % ROUND}
                # every instruction kind over the launch ALONE on the device: the stream as a whole against the issue rate
                if kernel_ms_alone and kernel_ms_alone.get("trace", 0.0) > 0:
                    out["roofline"]["valu_issue"]["all_kinds_alone"] = all_kinds_issue(name, kname, kernel_ms_alone["trace"])
                    out["roofline"]["valu_issue"]["valu_plus_salu_alone"] = valu_plus_salu_issue(name, kname, kernel_ms_alone["trace"])
                # ... and of the whole frame: launches of consecutive frames overlap (frames in flight), so what the chip's vector
                # pipes did per frame is every per-frame kernel's instructions over the FRAME time -- the roofline of the loop
                per_frame = measured_valu_per_frame(name)
                if per_frame:
                    fipc = per_frame / (ms_frame * 1e-3 * 2.4e9 * 1024)
                    out["roofline"]["valu_issue"]["frame"] = {"instructions_per_frame": int(per_frame), "achieved": round(fipc, 4), "peak": peak,
                                                              "frac": round(fipc / peak, 4), "frac_at_four_cycles": round(fipc / ISSUE_CEILING, 4),
                                                              "ms_per_frame": round(ms_frame, 5), "peak_is": "the trace kernel's ceiling (it issues most of the frame's vector instructions)"}
            algo_bytes = 4.0 * W * (y1 - y0) + 60.0 * len(tris)
            if kt > 0:
                out["roofline_hbm"] = {"bound": "hbm", "achieved": round(algo_bytes / (kt * 1e-3) / 1e9, 3), "peak": PEAK_HBM_GBS,
                                       "unit": "GB/s", "frac": round(algo_bytes / (kt * 1e-3) / 1e9 / PEAK_HBM_GBS, 5),
                                       "traffic": measured_traffic(name, [kname]) if world == 1 else None,
                                       "frame_traffic": measured_traffic(name, [""], per_frame=True) if world == 1 else None,
                                       "algorithmic_bytes": int(algo_bytes),
                                       "note": "algorithmic bytes = 4*W*rows framebuffer + 60*N triangle list; far below the HBM roof by "
                                               "construction; frame_traffic = every per-frame kernel (the PMC pass saw it more than twice), weighted by its launches per frame"}
        else:
            out.update({
                "metric": "frames/s (rasteriser)", "unit": "frames/s", "dtype": "f32", "value": round(nframes / dt, 3),
                "config": {"workload": name, "scene": "cornell-30", "triangles": int(len(tris)), "visible_triangles": int((culled0 == 0).sum()),
                           "width": W, "height": H, "lights": 1, "dof_kernel": dof, "covered_pixels": covered,
                           "parallelism": ("bands%d+%s" % (world, "rccl-p2p-gather(libmirt)" if native else "torch-gather")) if world > 1 else "1gpu"},
            })
            band_px = W * (y1 - y0)
            cov = covered if covered is not None else band_px
            tr = kernel_ms.get("raster_resolve", 0.0)
            mode_txt = "hipEvents around the launch on its own stream, %d frame(s) in flight, %s camera%s" % (in_flight, "moving" if moving else "static", "; the frame before the last of 6 queued back to back (overlapped on both sides)" if overlapped else "")
            small = len(tris) <= 64 and os.environ.get("MIRT_RASTER_SMALL", "1") != "0"
            if small and tr > 0:
                # Scenes of <= 64 triangles (the reference's Cornell box): k_raster_small keeps the depth test in registers and
                # writes 4 B per pixel; what it does is the reference's per-pixel arithmetic -- the winning fragment's interpolation
                # (Bresenham :661-662, 8 flop) and PixelShader (:549-589: 3 divisions by zinv, 15 for the inverse rotation, 3 + 9 + 1
                # camera / distance / focal, 41 per light incl. 3 divisions and 2 square roots, 9 for the colour, 12 to pack) =
                # 109 flop per covered pixel with one light, a division or a square root counted as ONE.  VALU-bound, no contraction.
                flop = RASTER_FLOP_PER_PIXEL * cov
                ach = flop / (tr * 1e-3) / 1e12
                out["roofline"] = {"bound": "valu", "kernel": "k_raster_small", "achieved": round(ach, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                                   "frac": round(ach / PEAK_FP32_TFLOPS, 4), "peak_reachable": PEAK_REACHABLE_TFLOPS,
                                   "frac_of_reachable": round(ach / PEAK_REACHABLE_TFLOPS["no_fma_packed"], 4),
                                   "primary_fraction": "valu_issue (a division is 11 and a square root 15 instructions on this chip, counted as one flop "
                                                       "each in `achieved`; issue slots count what the vector pipes do)",
                                   "traffic": measured_traffic(name, ["k_raster_small"]) if world == 1 else None,
                                   "traffic_source": "profiles/%s_hbm_traffic.json (rocprofv3 PMC, bytes per launch)" % ROUND,
                                   "algorithmic_flop": int(flop), "covered_pixels": int(cov), "kernel_ms": round(tr, 5), "kernel_ms_mode": mode_txt,
                                   "frame": {"achieved": round(flop / (ms_frame * 1e-3) / 1e12, 3), "frac": round(flop / (ms_frame * 1e-3) / 1e12 / PEAK_FP32_TFLOPS, 4),
                                             "ms_per_frame": round(ms_frame, 5)},
                                   "note": "FP32 VALU-bound: 109 flop per covered pixel as written in the reference (interpolation + PixelShader, one "
                                           "light), shaded once per pixel for the fragment that wins the depth test"}
                if kernel_ms_alone and kernel_ms_alone.get("raster_resolve", 0.0) > 0:
                    ka = kernel_ms_alone["raster_resolve"]
                    out["roofline"]["alone"] = {"kernel_ms": round(ka, 5), "achieved": round(flop / (ka * 1e-3) / 1e12, 3),
                                                "frac": round(flop / (ka * 1e-3) / 1e12 / PEAK_FP32_TFLOPS, 4)}
                insts = measured_valu_instructions(name, "k_raster_small") if world == 1 else None
                if insts:
                    ipc = insts / (tr * 1e-3 * 2.4e9 * 1024)
                    peak = issue_ceiling("k_raster_small")
                    out["roofline"]["valu_issue"] = {"instructions_per_launch": int(insts), "achieved": round(ipc, 4), "peak": peak,
                                                     "peak_is": "this kernel's instruction classes at their measured issue cost (tools/issue_mix.py; 0.24 = every instruction at four cycles)",
                                                     "unit": "wave-instr/clk/SIMD", "frac": round(ipc / peak, 4), "frac_at_four_cycles": round(ipc / ISSUE_CEILING, 4),
                                                     "lane_slots_per_pixel": round(insts * 64.0 / band_px, 1),
                                                     "source": "profiles/%s_pmc_issue.json (SQ_INSTS_VALU)" % ROUND}
                    if kernel_ms_alone and kernel_ms_alone.get("raster_resolve", 0.0) > 0:
                        out["roofline"]["valu_issue"]["all_kinds_alone"] = all_kinds_issue(name, "k_raster_small", kernel_ms_alone["raster_resolve"])
                        out["roofline"]["valu_issue"]["valu_plus_salu_alone"] = valu_plus_salu_issue(name, "k_raster_small", kernel_ms_alone["raster_resolve"])
                    per_frame = measured_valu_per_frame(name)
                    if per_frame:
                        fipc = per_frame / (ms_frame * 1e-3 * 2.4e9 * 1024)
                        out["roofline"]["valu_issue"]["frame"] = {"instructions_per_frame": int(per_frame), "achieved": round(fipc, 4), "peak": peak,
                                                                  "frac": round(fipc / peak, 4), "frac_at_four_cycles": round(fipc / ISSUE_CEILING, 4), "ms_per_frame": round(ms_frame, 5)}
                algo_bytes = 4.0 * band_px + 60.0 * len(tris)
                out["roofline_hbm"] = {"bound": "hbm", "achieved": round(algo_bytes / (tr * 1e-3) / 1e9, 3), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                       "frac": round(algo_bytes / (tr * 1e-3) / 1e9 / PEAK_HBM_GBS, 5),
                                       "traffic": measured_traffic(name, ["k_raster_small"]) if world == 1 else None,
                                       "frame_traffic": measured_traffic(name, ["k_cull", "k_raster"], per_frame=True) if world == 1 else None,
                                       "algorithmic_bytes": int(algo_bytes),
                                       "note": "4 B per pixel written + the triangle list; no depth-key buffer on this path"}
            elif tr > 0:
                # Larger scenes: k_raster_resolve reads the 8-byte depth key of every pixel and writes the XRGB word; it also
                # re-zeroes the keys it consumed (8 more bytes per COVERED pixel), which replaced the per-frame clear.
                rb = 12.0 * band_px + 8.0 * cov
                out["roofline"] = {"bound": "hbm", "kernel": "k_raster_resolve", "achieved": round(rb / (tr * 1e-3) / 1e9, 3),
                                   "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(rb / (tr * 1e-3) / 1e9 / PEAK_HBM_GBS, 5),
                                   "traffic": measured_traffic(name, ["k_raster_resolve"]) if world == 1 else None,
                                   "traffic_source": "profiles/%s_hbm_traffic.json (rocprofv3 PMC, bytes per launch)" % ROUND,
                                   "algorithmic_bytes": int(rb), "kernel_ms": round(tr, 5),
                                   "alone": (None if not (kernel_ms_alone and kernel_ms_alone.get("raster_resolve", 0.0) > 0) else
                                             {"kernel_ms": round(kernel_ms_alone["raster_resolve"], 5),
                                              "achieved": round(rb / (kernel_ms_alone["raster_resolve"] * 1e-3) / 1e9, 3),
                                              "frac": round(rb / (kernel_ms_alone["raster_resolve"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 5)}),
                                   "kernel_ms_mode": mode_txt,
                                   "note": "12 B per pixel (8 B key read + 4 B XRGB write) + 8 B per covered pixel (key re-zeroed)"}
                # The whole frame: fragments x 8 + resolve (above), over the frame time of the timed loop
                frag = 1.5 * cov
                algo_bytes = 12.0 * px + 8.0 * cov + 8.0 * frag
                out["roofline_frame"] = {"bound": "hbm", "achieved": round(algo_bytes / (ms_frame * 1e-3) / 1e9, 3), "peak": PEAK_HBM_GBS,
                                         "unit": "GB/s", "frac": round(algo_bytes / (ms_frame * 1e-3) / 1e9 / PEAK_HBM_GBS, 5),
                                         "traffic": measured_traffic(name, ["k_cull", "k_raster"]) if world == 1 else None,
                                         "algorithmic_bytes": int(algo_bytes),
                                         "kernel_ms_sum": round(sum(kernel_ms.get(k, 0.0) for k in ("clear", "raster_setup", "raster_frag", "raster_resolve")), 5)}
        if dof and kernel_ms.get("dof", 0.0) > 0:
            # the depth-of-field pass: K*K taps x 3 channels x (multiply, add) per pixel, two per packed
            # instruction -- VALU-bound; its issue fraction from the committed PMC pass of this workload
            kd = kernel_ms["dof"]
            floor_insts = W * (y1 - y0) * dof * dof * 3.0 / 64.0          # packed wave-instructions per launch at two operations each
            dinsts = measured_valu_instructions(name, "k_dof_tile") if world == 1 else None
            out["dof"] = {"kernel": "k_dof_tile<%d>" % dof, "kernel_ms": round(kd, 5),
                          "kernel_ms_alone": round(kernel_ms_alone["dof"], 5) if kernel_ms_alone and kernel_ms_alone.get("dof") else None,
                          "packed_instruction_floor_per_launch": int(floor_insts),
                          "floor_ms_at_issue_ceiling": round(floor_insts / (ISSUE_CEILING * 2.4e9 * 1024) * 1e3, 5),
                          "valu_issue": None if not dinsts else {"instructions_per_launch": int(dinsts), "achieved": round(dinsts / (kd * 1e-3 * 2.4e9 * 1024), 4),
                                                                 "peak": issue_ceiling("k_dof_tile<8>"), "unit": "wave-instr/clk/SIMD",
                                                                 "frac": round(dinsts / (kd * 1e-3 * 2.4e9 * 1024) / issue_ceiling("k_dof_tile<8>"), 4),
                                                                 "source": "profiles/%s_pmc_issue.json (SQ_INSTS_VALU)" % ROUND}}
        if world == 1 and want_cpu:
            # deferred (main() runs it after every GPU measurement of the line): tens of seconds of host-only work let the GPU
            # drop its clocks, and the workload timed next would start on a cold device
            def cpu_leg(rot=mirt.rot_from_yaw(0.0, rot11)):
                r = cpu_baseline(kind, tris, culled0, W, H, cam, rot, focal, samples=soft_samples, jitter=soft_jitter, aa=aa, budget_s=12.0 if extras else 5.0)
                if dof:
                    r["sample"] += "; per-pixel path only, the depth-of-field blur is not part of the CPU sample"
                return r
            out["_cpu_leg"] = cpu_leg
    mirt.set_soft_shadows(1)
    mirt.set_antialiasing(1)
    mirt.set_depth_of_field(0)
    mirt.set_frames_in_flight(1)
    if world > 1:
        mirt.set_partition(0)
    return out


def host_path_rate(mirt, kind, views, mode, W, H, t_frame):
    """Frames per second through the host-surface entry points (what the reference's Draw() binds): render + the
    framebuffer's way into the caller's (SDL) surface, one frame after the other, synchronous per frame.  Measured for a
    pageable surface and for one the caller registered (mirt_surface_register: pinned + mapped, kernels store into it)."""
    n = int(max(4, min(200, 0.25 / max(t_frame * 10, 1e-5))))

    def rate(surf):
        if kind == "rt":
            call = lambda v: mirt.raytrace(v, LIGHT, INDIRECT, mode, want_rgb=False, want_index=False, xrgb=surf)        # noqa: E731
        else:
            call = lambda v: mirt.rasterise(v, LIGHT, INDIRECT, want_rgb=False, want_zinv=False, want_index=False, xrgb=surf)   # noqa: E731
        for i in range(3):
            call(views[i % len(views)])
        t0 = time.perf_counter()
        for i in range(n):
            call(views[i % len(views)])
        dt = (time.perf_counter() - t0) / n
        return {"frames_per_s": round(1.0 / dt, 2), "ms_per_frame": round(dt * 1e3, 4), "surface_GB_per_s": round(W * H * 4 / dt / 1e9, 2)}

    out = {"frames": n, "pageable": rate(np.zeros((H, W), np.uint32))}
    surf = np.zeros((H, W), np.uint32)
    try:
        mirt.surface_register(surf)
        out["registered"] = rate(surf)
        out["registered"]["link_frac"] = round(out["registered"]["surface_GB_per_s"] / 63.0, 3)
        mirt.surface_unregister(surf)
    except mirt.MirtError as e:
        out["registered"] = {"error": str(e)}
    # asynchronous frames (mirt_*_async): two registered surfaces in turn, two frames in flight, ONE sync at the end -- the loop
    # that presents one surface while the next is drawn; the copy of frame i overlaps the render of frame i + 1
    try:
        pair = np.zeros((2, H, W), np.uint32)
        mirt.surface_register(pair)
        was = 2                                             # the timed loop ran with two frames in flight; left that way below
        mirt.set_frames_in_flight(2)
        calls = [mirt.prepared_async(kind, v, LIGHT, INDIRECT, mode, pair[i & 1]) for i, v in enumerate(views[:16])]
        for i in range(4):
            calls[i % len(calls)]()
        mirt.sync()
        t0 = time.perf_counter()
        for i in range(n):
            calls[i % len(calls)]()
        mirt.sync()
        adt = (time.perf_counter() - t0) / n
        out["async"] = {"frames_per_s": round(1.0 / adt, 2), "ms_per_frame": round(adt * 1e3, 4), "surface_GB_per_s": round(W * H * 4 / adt / 1e9, 2),
                        "link_frac": round(W * H * 4 / adt / 1e9 / 63.0, 3), "frames_in_flight": was}
        mirt.surface_unregister(pair)
    except mirt.MirtError as e:
        out["async"] = {"error": str(e)}
    # the floor: a bare device-to-host copy of the same surface into pinned memory (no render), same sync per frame
    try:
        import torch
        src = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        dst = torch.empty((H, W), dtype=torch.int32, pin_memory=True)
        for i in range(3):
            dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize()
        cdt = (time.perf_counter() - t0) / n
        out["bare_pinned_copy"] = {"ms_per_frame": round(cdt * 1e3, 4), "surface_GB_per_s": round(W * H * 4 / cdt / 1e9, 2)}
    except Exception as e:                                  # noqa: BLE001  (a measurement beside the result, never fatal)
        out["bare_pinned_copy"] = {"error": str(e)}
    best = min((out[k] for k in ("pageable", "registered") if "ms_per_frame" in out.get(k, {})), key=lambda r: r["ms_per_frame"])
    if "ms_per_frame" in out["bare_pinned_copy"]:
        out["copy_floor_frac"] = round(out["bare_pinned_copy"]["ms_per_frame"] / best["ms_per_frame"], 3)
        if "ms_per_frame" in out.get("async", {}):
            out["async"]["copy_floor_frac"] = round(out["bare_pinned_copy"]["ms_per_frame"] / out["async"]["ms_per_frame"], 3)
    out.update({"frames_per_s": best["frames_per_s"], "ms_per_frame": best["ms_per_frame"],
                "note": "mirt_raytrace / mirt_rasterise into a host surface (the SDL surface of the reference): render + delivery, "
                        "synchronous per frame; registered = mirt_surface_register'ed surface; PCIe Gen5 x16 is 63 GB/s (link_frac); "
                        "bare_pinned_copy = the same bytes copied device-to-host with no render at all, copy_floor_frac = that floor / our frame; "
                        "async = mirt_*_async into two registered surfaces in turn, two frames in flight, one sync at the end (the copy of a "
                        "frame overlaps the render of the next)"})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="auto", choices=["auto", "brute", "binned"])
    ap.add_argument("--static-camera", action="store_true", help="one fixed view for every frame (the sizing caches always hit)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sub-results", action="store_true")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` launches its own ranks: one process per GPU under torch.distributed.run, started HERE, before
        # this process has touched torch or the GPU (a process that has initialised HIP must never be replaced or forked from);
        # the parent only waits and hands on the exit code -- non-zero if any rank died.
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    env = Env(args)
    steps = args.steps if args.steps is not None else 20
    warmup = args.warmup if args.warmup is not None else 3
    name = args.workload or DEFAULT_WORKLOAD
    # the timed region lasts a second at least (an activity sampler beside the run sees it); MIRT_BENCH_TARGET_S overrides
    target = float(os.environ.get("MIRT_BENCH_TARGET_S", "1.0")) if args.mode != "brute" else 0.0
    heavy_brute = args.mode == "brute" and WORKLOADS[name][1][0] == "soup"
    out = run_workload(env, name, steps, warmup, args.mode, not args.static_camera, not args.no_cpu_baseline, target_s=target, quick=heavy_brute)
    if args.workload is None and not args.no_sub_results:
        subs = {}
        keep = ("metric", "value", "unit", "ms_per_frame", "frames_per_s", "frames_per_step", "steps", "timed_region_s",
                "kernel_ms_rank0", "kernel_ms_alone_rank0", "config", "roofline", "_cpu_leg")
        for sub in (SUB_RESULTS if env.world == 1 else SUB_RESULTS_SHARDED):
            # (no CPU leg for the 1 M-triangle frame: one 8K row of it is seconds of brute force on 64 cores)
            r = run_workload(env, sub, max(4, steps // 2), 1, "auto", not args.static_camera, not args.no_cpu_baseline and sub != "soup1m8k", target_s=min(target, 0.25), extras=False)
            if r is not None:
                subs[sub] = {k: r[k] for k in keep if k in r}
        if env.world == 1 and name == "soup100k" and args.mode != "brute":
            # BASELINE configs[2] as worded -- "LDS-tiled brute-force intersect": ONE frame of the default workload through k_rt_brute
            # (152 ms), the one kernel whose executed work IS the reference's rays x triangles x 60 flop, so its roofline fraction is
            # the full-work one; the binned line above renders the same frame bit for bit in a 2 000th of the time
            r = run_workload(env, "soup100k", 1, 0, "brute", not args.static_camera, False, target_s=0.0, extras=False, quick=True)
            if r is not None:
                subs["soup100k_brute"] = {k: r[k] for k in keep if k in r}
        if out is not None:
            out["sub_results"] = subs
    # the CPU legs, after all GPU timing
    for rec in ([out] if out is not None else []) + list((out or {}).get("sub_results", {}).values()):
        leg = rec.pop("_cpu_leg", None)
        if leg is not None:
            rec["cpu_baseline"] = leg()
    if env.rank == 0:
        print(json.dumps(out), flush=True)
    if env.world > 1:
        env.dist.barrier()
        env.dist.destroy_process_group()
    env.mirt.shutdown()


if __name__ == "__main__":
    main()
