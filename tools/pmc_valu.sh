#!/usr/bin/env bash
# tools/pmc_valu.sh <tag> <bench args...> -- issue-side counters (VALU / LDS busy, waits, bank conflicts) of every kernel
# of a short bench.py run; kernel-trace only, counters in their own passes.
set -uo pipefail
tag="$1"; shift
out="gpurun_out/pmcv_$tag"
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --no-cpu-baseline "$@" > "$out/bench_trace.json" 2> "$out/trace.err"
echo "trace rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$out/pmc1" -- python3 bench.py --no-cpu-baseline "$@" > "$out/bench_pmc1.json" 2> "$out/pmc1.err"
echo "pmc1 rc=$?"
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_IFETCH SQ_INST_CYCLES_VMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc2" -- python3 bench.py --no-cpu-baseline "$@" > "$out/bench_pmc2.json" 2> "$out/pmc2.err"
echo "pmc2 rc=$?"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:1500])
for sub in ("pmc1", "pmc2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(out + "/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(sub, k, {c: round(v / max(1, cnt[(k, c)]), 1) for c, v in d.items()})
PY
