#!/usr/bin/env bash
set -uo pipefail
export TMPDIR=/tmp
out=gpurun_out/r04g
mkdir -p $out
for lazy in 0 1; do
  MIRT_LAZY_GEO=$lazy timeout -k 10 200 python tools/band_prof.py 0 4320 10 move > $out/full_lazy$lazy.txt 2>&1; echo "lazy=$lazy: $(grep rows $out/full_lazy$lazy.txt)"
  MIRT_LAZY_GEO=$lazy MIRT_BENCH_TARGET_S=0.3 timeout -k 10 300 python bench.py --workload soup1m8k --no-cpu-baseline --steps 5 --warmup 2 > $out/bench_soup1m8k_lazy$lazy.json 2> $out/err.txt; python3 -c "
import json;d=json.load(open('$out/bench_soup1m8k_lazy$lazy.json'));print('lazy=$lazy', {k:d[k] for k in ('value','ms_per_frame','kernel_ms_rank0','kernel_ms_alone_rank0')})"
  MIRT_LAZY_GEO=$lazy MIRT_BENCH_TARGET_S=0.3 timeout -k 10 300 python bench.py --workload soup100k --no-cpu-baseline > $out/bench_soup100k_lazy$lazy.json 2> $out/err.txt; python3 -c "
import json;d=json.load(open('$out/bench_soup100k_lazy$lazy.json'));print('lazy=$lazy', {k:d[k] for k in ('value','ms_per_frame','kernel_ms_rank0','kernel_ms_alone_rank0')})"
done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "environment_variant or baseline" > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.txt
timeout -k 10 300 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -x -q > $out/pytest2.txt 2>&1; echo "pytest2 rc=$?"; tail -3 $out/pytest2.txt
echo done
