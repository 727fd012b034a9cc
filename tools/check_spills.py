#!/usr/bin/env python3
"""tools/check_spills.py -- compiles every kernel source of the library for gfx950 (device side only, the Makefile's flags) and lists
each kernel's VGPR count, occupancy and scratch bytes (and which kernels sit within eight registers of another wave per SIMD); exits non-zero if any kernel spills to scratch (private segment != 0).
__graft_entry__.build() runs it: a spill in a hot loop is a silent 2-3x."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cpp-raytracer-rasterizer_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-x", "hip", "--cuda-device-only", "-S"]
bad = []
rows = []
with tempfile.TemporaryDirectory() as tmp:
    for name in sorted(os.listdir(CSRC)):
        if not name.endswith(".hip"):
            continue
        out = os.path.join(tmp, name + ".s")
        cc = subprocess.run(["hipcc"] + FLAGS + ["-c", os.path.join(CSRC, name), "-o", out], capture_output=True, text=True)
        if cc.returncode != 0:
            sys.stderr.write(cc.stderr)
            sys.exit("tools/check_spills.py: hipcc failed on %s (rc %d)" % (name, cc.returncode))
        text = open(out).read()
        # the per-kernel resource comments the backend emits: "; Kernel ... " blocks end with NumVgprs / ScratchSize / Occupancy
        for m in re.finditer(r"^\s*\.set (\S+)\.uses_flat_scratch.*?; NumVgprs: (\d+).*?; ScratchSize: (\d+).*?; Occupancy: (\d+)", text, re.S | re.M):
            kern = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0]
            rows.append((name, kern, int(m.group(2)), int(m.group(3)), int(m.group(4))))
            if int(m.group(3)):
                bad.append(rows[-1])
def waves_by_registers(vgprs):
    """Waves per SIMD the register file allows: 512 VGPRs, handed out in blocks of eight (the backend's `Occupancy` also knows the LDS and
    the kernel's waves-per-eu hint; this is the registers' own limit)."""
    return min(8, 512 // (((max(vgprs, 1) + 7) // 8) * 8))


for r in rows:
    # how close the kernel is to the next wave per SIMD: round 4 found k_rt_trace2 two registers and k_rt_tile2 twelve above a boundary
    # that was worth 5 % and 15 % of their frames
    w = waves_by_registers(r[2])
    over = r[2] - (512 // (w + 1)) // 8 * 8 if w < 8 else 0
    hint = "   (%d VGPRs above %d waves per SIMD)" % (over, w + 1) if 0 < over <= 8 else ""
    print("%-22s %-44s vgprs %3d scratch %4d occupancy %d%s" % (r + (hint,)))
# the gate must not pass vacuously: should the backend's resource comments change shape, the pattern above matches nothing
EXPECT = ("k_rt_trace2", "k_rt_tile2", "k_rt_brute", "k_bin_pairs", "k_raster_small", "k_raster_resolve", "k_dof_tile", "k_bs_local")
missing = [k for k in EXPECT if not any(k in r[1] for r in rows)]
if missing or len(rows) < 20:
    sys.exit("tools/check_spills.py: resource summaries found for %d kernels only; missing %s -- the pattern no longer matches the backend's output"
             % (len(rows), missing))
if bad:
    print("SPILLS:", [(b[1], b[3]) for b in bad])
    sys.exit(1)
print("no kernel spills to scratch (%d kernels)" % len(rows))
