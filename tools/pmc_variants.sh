#!/usr/bin/env bash
# tools/pmc_variants.sh <kernel substring|all> <profiled script> [its args ...] -- libs <variant.so ...>
# (GPU box) issue-side counters of ONE kernel for the shipped library and for alternative builds of it (tools/build_variant.sh):
# e.g.  tools/pmc_variants.sh k_rt_tile2 tools/trace_prof.py cornell -- build/variants/libmirt_x.so
set -uo pipefail
export TMPDIR=/tmp
kernel="$1"; shift
args=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
[ $# -gt 0 ] && shift
for lib in "" "$@"; do
  out=gpurun_out/pmcv_tmp; rm -rf $out; mkdir -p $out
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc1 -- python3 "${args[@]}" $lib > $out/p1.txt 2>&1
  python3 - $out "$kernel" "${lib:-shipped}" <<'PY'
import csv, glob, sys, collections
out, kernel, tag = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/pmc1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if "mirt" not in name or (kernel != "all" and kernel not in name): continue
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(name, r["Counter_Name"])] += 1
for name, d in acc.items():
    print(tag.split("/")[-1], name, {c: round(v / max(1, cnt[(name, c)])) for c, v in sorted(d.items())}, "launches", cnt[(name, "SQ_WAVES")], flush=True)
PY
  rm -rf $out
done
