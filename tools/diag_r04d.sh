#!/usr/bin/env bash
set -uo pipefail
export TMPDIR=/tmp
out=gpurun_out/r04d
mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -8 $out/pytest.txt
timeout -k 10 300 python tools/fuzz_binned.py 2000 150 6 > $out/fuzz_binned.txt 2>&1; echo "fuzz binned rc=$?"; tail -3 $out/fuzz_binned.txt
timeout -k 10 300 python tools/fuzz_sequence.py 2000 100 > $out/fuzz_sequence.txt 2>&1; echo "fuzz sequence rc=$?"; tail -3 $out/fuzz_sequence.txt
timeout -k 10 300 python tools/band_cost.py 8 both > $out/band_cost.txt 2>&1; echo "band cost rc=$?"; cat $out/band_cost.txt
timeout -k 10 200 python tools/moving_light.py > $out/moving_light.txt 2>&1; echo "moving light rc=$?"; cat $out/moving_light.txt
MIRT_BENCH_TARGET_S=0.3 timeout -k 10 300 python bench.py --workload soup100k --no-cpu-baseline > $out/bench_soup100k.json 2> $out/bench_soup100k.err; echo "bench rc=$?"; python3 -c "
import json;d=json.load(open('$out/bench_soup100k.json'));print({k:d[k] for k in ('value','ms_per_frame','kernel_ms_rank0','kernel_ms_alone_rank0','static_camera')})"
MIRT_BENCH_TARGET_S=0.3 timeout -k 10 300 python bench.py --workload soup1m8k --no-cpu-baseline --steps 5 --warmup 2 > $out/bench_soup1m8k.json 2> $out/bench_soup1m8k.err; echo "bench 8k rc=$?"; python3 -c "
import json;d=json.load(open('$out/bench_soup1m8k.json'));print({k:d[k] for k in ('value','ms_per_frame','kernel_ms_rank0','kernel_ms_alone_rank0','static_camera')})"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_sq -- python3 tools/trace_prof.py > $out/pmc_sq.txt 2>&1; echo "pmc sq rc=$?"
python3 - $out <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:60]
        if "mirt" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k, {c: round(v / max(1, cnt[(k, c)]), 1) for c, v in d.items()}, "launches", max(cnt[(k, c)] for c in d))
PY
find $out -name "*counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
echo done
