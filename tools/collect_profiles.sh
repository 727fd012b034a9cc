#!/usr/bin/env bash
# tools/collect_profiles.sh pmc|fuzz|bench -- (GPU box) every record kept under profiles/ for a round, in three gpurun calls:
#   pmc    rocprofv3 kernel stats, the HBM-traffic PMC passes and the issue-side PMC passes of the default workloads;
#   fuzz   the microbenchmarks, the fuzzers and the moving-light run;
#   bench  the bench lines (with cpu_baseline) -- AFTER `python tools/store_profiles.py <tag>` has put the PMC summaries of
#          the first call into profiles/, which is where bench.py reads `roofline.traffic` from.
# Results land in gpurun_out/; tools/store_profiles.py copies the summaries into profiles/ (in the build container).
set -uo pipefail
stage="${1:-pmc}"
shift || true
if [ "$stage" = pmc ]; then
  # (optionally: tools/collect_profiles.sh pmc <workload> ... -- a gpurun call is at most 20 minutes)
  [ $# -gt 0 ] || set -- soup100k cornell1080 raster4k soup1m8k raster4kdof8 cornell1080dof8
  # the digest of the device code the counters are about to count: taken HERE, on the box that runs them, and only copied later
  stamp=$(python3 -c "import bench; print(bench.csrc_digest())")
  export MIRT_BENCH_TARGET_S=0.1            # (a profiled run is about launches, not about a long timed region)
  for t in "$@"; do
    steps=20; [ $t = soup1m8k ] && steps=4
    tools/prof.sh $t --workload $t --steps $steps --warmup 3 > /dev/null 2>&1; echo "trace $t rc=$?"
    tools/pmc_hbm.sh $t --workload $t --steps $((steps / 2)) --warmup 2 > /dev/null 2>&1
    python tools/pmc_summary.py gpurun_out/pmc_$t > gpurun_out/pmc_$t/summary.json
    echo $stamp > gpurun_out/pmc_$t/csrc_sha16.txt
    tools/pmc_valu.sh $t --workload $t --steps $((steps / 2)) --warmup 2 > gpurun_out/pmcv_$t.txt 2>&1
    python tools/pmc_issue_summary.py gpurun_out/pmcv_$t > gpurun_out/pmcv_$t/summary.json; echo "pmc $t rc=$?"
    echo $stamp > gpurun_out/pmcv_$t/csrc_sha16.txt
    # the per-dispatch tables are tens of MiB per pass (gpurun merges at most 64 MiB back): the summaries are what is kept
    find gpurun_out/prof_$t gpurun_out/pmc_$t gpurun_out/pmcv_$t -name "*counter_collection.csv" -delete -o -name "*kernel_trace.csv" -delete
  done
elif [ "$stage" = fuzz ]; then
  tools/ubench > gpurun_out/ubench.txt 2>&1
  tools/div2check > gpurun_out/divcheck.txt 2>&1; tools/div3check >> gpurun_out/divcheck.txt 2>&1; echo "division checks rc=$?"
  tools/edgebench > gpurun_out/edgebench.txt 2>&1
  python tools/moving_light.py > gpurun_out/moving_light.txt 2>&1; echo "moving light rc=$?"
  python tools/fuzz_binned.py 0 300 6 > gpurun_out/fuzz_binned_vs_brute.txt 2>&1; echo "fuzz rc=$?"
  python tools/fuzz_small.py 0 3000 > gpurun_out/fuzz_small_scenes_vs_oracle.txt 2>&1; echo "fuzz small rc=$?"
  python tools/fuzz_sequence.py 0 200 > gpurun_out/fuzz_call_sequences.txt 2>&1; echo "fuzz sequences rc=$?"
  python tools/fuzz_raster_sequence.py 0 200 > gpurun_out/fuzz_raster_call_sequences.txt 2>&1; echo "fuzz raster sequences rc=$?"
else
  python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench default rc=$?"
  for w in soup100k cornell1080 raster4k cornell500 cornell1080soft16 cornell1080aa3 cornell1080dof8 raster4kdof8; do
    python bench.py --workload $w > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err; echo "bench $w rc=$?"
  done
  python bench.py --workload soup1m8k --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_soup1m8k.json 2>/dev/null; echo "bench soup1m8k rc=$?"
  python bench.py --workload soup100k --mode brute --steps 2 --warmup 1 --no-cpu-baseline --static-camera > gpurun_out/bench_soup100k_brute.json 2>/dev/null; echo "bench soup100k brute rc=$?"
fi
echo done
