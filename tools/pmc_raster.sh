#!/usr/bin/env bash
# tools/pmc_raster.sh <tag> -- issue-side counters of the rasteriser's kernels (tools/raster_prof.py), counters in their own passes
set -uo pipefail
tag="$1"; shift
out="gpurun_out/pmcr_$tag"
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$out/pmc1" -- python3 tools/raster_prof.py "$@" > "$out/p1.txt" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc2" -- python3 tools/raster_prof.py "$@" > "$out/p2.txt" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("pmc1", "pmc2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(out + "/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:36]
            if "mirt" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(sub, k, {c: round(v / max(1, cnt[(k, c)])) for c, v in d.items()})
PY
