#!/usr/bin/env bash
set -uo pipefail
export TMPDIR=/tmp
out=gpurun_out/r04h
mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_golden_ref_render.py -m gpu -x -q -k "raster" > $out/pytest.txt 2>&1; echo "pytest raster rc=$?"; tail -3 $out/pytest.txt
timeout -k 10 300 python tools/fuzz_small.py 3000 400 > $out/fuzz_small.txt 2>&1; echo "fuzz small rc=$?"; tail -2 $out/fuzz_small.txt
timeout -k 10 300 python tools/fuzz_raster_sequence.py 3000 60 > $out/fuzz_raster.txt 2>&1; echo "fuzz raster rc=$?"; tail -2 $out/fuzz_raster.txt
for per in 4 5 6 8; do
  echo "prefetch, $per workgroups per CU: $(MIRT_SMALL_WGS_PER_CU=$per python tools/raster_variant.py 2>&1 | tail -1)"
  echo "no prefetch, $per workgroups per CU: $(MIRT_SMALL_WGS_PER_CU=$per python tools/raster_variant.py cpp-raytracer-rasterizer_amd/build/variants/libmirt_nopf.so 2>&1 | tail -1)"
done
for per in 4 6; do
MIRT_SMALL_WGS_PER_CU=$per MIRT_BENCH_TARGET_S=0.3 timeout -k 10 300 python bench.py --workload raster4k --no-cpu-baseline > $out/bench_raster4k_$per.json 2> $out/err.txt; python3 -c "
import json;d=json.load(open('$out/bench_raster4k_$per.json'));print('per $per', {k:d[k] for k in ('value','ms_per_frame','kernel_ms_rank0','kernel_ms_alone_rank0')})"
done
echo done
for lib in "" cpp-raytracer-rasterizer_amd/build/variants/libmirt_w5.so; do
  echo "trace variant [$lib] 100k: $(python tools/soup_variant.py $lib 2>&1 | tail -2 | tr '\n' ' ')"
  echo "trace variant [$lib] 8K: $(python tools/band_prof.py 0 4320 8 move $lib 2>&1 | grep rows)"
done
