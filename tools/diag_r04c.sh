#!/usr/bin/env bash
set -uo pipefail
export TMPDIR=/tmp
out=gpurun_out/r04c
mkdir -p $out
for band in "0 540" "1620 2160" "0 4320"; do
  tag=$(echo $band | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/band_$tag -- python3 tools/band_prof.py $band 12 move > $out/band_$tag.txt 2>&1; echo "band $band rc=$?"
  find $out/band_$tag -name "*kernel_stats.csv" | head -1 | xargs -r cat > $out/band_${tag}_kernel_stats.csv
  find $out/band_$tag -name "*kernel_trace.csv" -delete
  python3 - $out/band_${tag}_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    print("  %-34s calls %3s avg %9.1f us  min %9.1f  max %9.1f" % (n[:34], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  cat $out/band_$tag.txt | grep rows
done
timeout -k 10 300 python tools/band_cost.py 8 both > $out/band_cost.txt 2>&1; echo "band cost rc=$?"; cat $out/band_cost.txt
MIRT_BENCH_TARGET_S=0.2 timeout -k 10 300 python bench.py --workload soup100k --no-cpu-baseline > $out/bench_soup100k.json 2> $out/bench_soup100k.err; echo "bench rc=$?"; python3 -c "
import json;d=json.load(open('$out/bench_soup100k.json'));print({k:d[k] for k in ('value','ms_per_frame','kernel_ms_rank0','kernel_ms_alone_rank0','static_camera')})"
find $out -name "*agent_info.csv" -delete
echo done
