#!/usr/bin/env bash
# tools/dof_ab.sh [libmirt variant.so] -- (GPU box) k_dof_tile<8> at 4K: its time alone (tools/dof_prof.py, hipEvents), then its fetched bytes and
# executed vector instructions per launch (rocprofv3 --pmc, one counter per pass), for the shipped library or an A/B build.
set -uo pipefail
export TMPDIR=/tmp
V="${1:-}"
tag="${2:-new}"
python3 tools/dof_prof.py $V > gpurun_out/dof_$tag.txt 2>&1 && python3 tools/dof_prof.py $V >> gpurun_out/dof_$tag.txt 2>&1
cat gpurun_out/dof_$tag.txt
for ctr in FETCH_SIZE SQ_INSTS_VALU SQ_INSTS_SALU; do
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/dofpmc_${tag}_$ctr -- python3 tools/dof_prof.py $V > /dev/null 2> gpurun_out/dofpmc_${tag}_$ctr.err || echo "pmc $ctr failed"
done
python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
for fn in sorted(glob.glob("gpurun_out/dofpmc_%s_*/**/*counter_collection.csv" % tag, recursive=True)):
    acc = {}
    for r in csv.DictReader(open(fn)):
        if "k_dof_tile" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print(tag, {k: (round(sum(v) / len(v), 1), len(v)) for k, v in acc.items()})
PY
