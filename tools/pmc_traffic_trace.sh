#!/usr/bin/env bash
# tools/pmc_traffic_trace.sh <tag> [nolight] -- fabric-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and L2 hit counters of
# the binned frame's kernels (tools/trace_prof.py: 100 k soup at 1080p, static camera)
set -uo pipefail
tag="$1"; shift
out="gpurun_out/pmctt_$tag"
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 tools/trace_prof.py "$@" > "$out/f.txt" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 tools/trace_prof.py "$@" > "$out/w.txt" 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/l2" -- python3 tools/trace_prof.py "$@" > "$out/l2.txt" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for sub in ("fetch", "write", "l2"):
    for f in glob.glob(out + "/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:36]
            if "mirt" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    v = {c: d[c] / max(1, cnt[(k, c)]) for c in d}
    print(k, "fetch MB %.1f write MB %.1f  L2 hit %.0f miss %.0f (hit rate %.2f)" % (v.get("FETCH_SIZE", 0) * 2 * 1024 / 1e6, v.get("WRITE_SIZE", 0) * 1024 / 1e6,
          v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0), v.get("TCC_HIT_sum", 0) / max(1.0, v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0))))
PY
