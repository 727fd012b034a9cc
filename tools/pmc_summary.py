"""tools/pmc_summary.py <gpurun_out/pmc_TAG> -- per-kernel HBM bytes per launch from the two PMC passes of
tools/pmc_hbm.sh.  FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 counts 64 B per 128-B request
of a wide coalesced read, MI355X_MICROARCH.md section HBM).  Prints JSON: {kernel: {launches, fetch_bytes, write_bytes}} for
EVERY kernel of the run (the library's own, the runtime's fill / copy kernels, torch's) -- whatever touches HBM in a frame."""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
out = collections.defaultdict(lambda: {"launches": 0, "fetch_bytes": 0.0, "write_bytes": 0.0})
for which, key, mul in (("fetch", "fetch_bytes", 2.0), ("write", "write_bytes", 1.0)):
    files = glob.glob("%s/%s/*/*counter_collection.csv" % (root, which))
    n = collections.Counter()
    tot = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:80]
            tot[k] += float(r["Counter_Value"]) * 1024.0 * mul
            n[k] += 1
    for k in tot:
        out[k][key] = tot[k] / max(n[k], 1)
        out[k]["launches"] = max(out[k]["launches"], n[k])
print(json.dumps({k: {a: (round(b) if a != "launches" else b) for a, b in v.items()} for k, v in out.items()}, indent=1))
