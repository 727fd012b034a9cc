#!/usr/bin/env bash
# tools/final_collect.sh [budget seconds] -- (GPU box) ONE gpurun call for the end of a round, when the GPU minutes left do not
# allow tools/collect_profiles.sh's three: the full `-m gpu` suite, then counters and bench lines workload by workload in the
# order that matters (the driver's default line first), each stage skipped once the budget is nearly spent.  The PMC summaries
# go into the box's own profiles/ (tools/store_profiles.py) before the bench lines read them; gpurun_out/ carries everything
# back, and the same store step is repeated in the build container.
set -uo pipefail
budget="${1:-1000}"
tag="${2:-r04}"
t0=$(date +%s)
left() { echo $((budget - ($(date +%s) - t0))); }
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x -p no:cacheprovider > gpurun_out/pytest_final.txt 2>&1; rc=$?
echo "pytest rc=$rc ($(tail -1 gpurun_out/pytest_final.txt)) left $(left) s"
[ $rc -eq 0 ] || exit $rc
for t in soup100k cornell1080; do
  bash tools/collect_profiles.sh pmc $t > gpurun_out/collect_$t.txt 2>&1; echo "pmc $t rc=$? left $(left) s"
done
python tools/store_profiles.py $tag > /dev/null
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench default rc=$? left $(left) s"
for t in raster4k soup1m8k cornell1080dof8 raster4kdof8; do
  need=150; [ $t = soup1m8k ] && need=260
  if [ $(left) -lt $need ]; then echo "pmc $t skipped (left $(left) s)"; continue; fi
  bash tools/collect_profiles.sh pmc $t > gpurun_out/collect_$t.txt 2>&1; echo "pmc $t rc=$? left $(left) s"
done
python tools/store_profiles.py $tag > /dev/null
for w in cornell1080 raster4k soup100k cornell1080aa3 cornell1080soft16 cornell1080dof8 raster4kdof8 cornell500; do
  if [ $(left) -lt 60 ]; then echo "bench $w skipped (left $(left) s)"; continue; fi
  python bench.py --workload $w > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err; echo "bench $w rc=$? left $(left) s"
done
if [ $(left) -ge 60 ]; then
  python bench.py --workload soup1m8k --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_soup1m8k.json 2>/dev/null; echo "bench soup1m8k rc=$? left $(left) s"
fi
echo done
