#!/usr/bin/env python3
"""tools/timeline.py <kernel_trace.csv> [frames] -- the launches of the last `frames` frames of a rocprofv3 --kernel-trace run as a
timeline: start (us, relative), duration, gap to the previous kernel's end.  A frame ends with its trace / raster kernel."""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ends = [i for i, r in enumerate(rows) if "k_rt_trace2" in r["Kernel_Name"] or "k_raster_small" in r["Kernel_Name"] or "k_raster_resolve" in r["Kernel_Name"] or "k_rt_tile2" in r["Kernel_Name"]]
if len(ends) > frames:
    rows = rows[ends[-frames - 1] + 1:]
t0 = int(rows[0]["Start_Timestamp"])
prev = None
for r in rows:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("mirt::", "")
    print("%9.1f us  %-28s %8.1f us   gap %7.1f" % ((a - t0) / 1e3, name[:28], (b - a) / 1e3, 0.0 if prev is None else (a - prev) / 1e3))
    prev = b
