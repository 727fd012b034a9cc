#!/usr/bin/env python3
"""tools/check_spills.py -- compiles every kernel source of the library for gfx950 (device side only, the Makefile's flags) and lists
each kernel's VGPR count, occupancy and scratch bytes; exits non-zero if any kernel spills to scratch (private segment != 0).
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
        subprocess.check_call(["hipcc"] + FLAGS + ["-c", os.path.join(CSRC, name), "-o", out], stderr=subprocess.DEVNULL)
        text = open(out).read()
        # the per-kernel resource comments the backend emits: "; Kernel ... " blocks end with NumVgprs / ScratchSize / Occupancy
        for m in re.finditer(r"^\s*\.set (\S+)\.uses_flat_scratch.*?; NumVgprs: (\d+).*?; ScratchSize: (\d+).*?; Occupancy: (\d+)", text, re.S | re.M):
            kern = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
            rows.append((name, kern, int(m.group(2)), int(m.group(3)), int(m.group(4))))
            if int(m.group(3)):
                bad.append(rows[-1])
for r in rows:
    print("%-22s %-44s vgprs %3d scratch %4d occupancy %d" % r)
if bad:
    print("SPILLS:", [(b[1], b[3]) for b in bad])
    sys.exit(1)
print("no kernel spills to scratch (%d kernels)" % len(rows))
