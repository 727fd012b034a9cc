#!/usr/bin/env bash
set -uo pipefail
export TMPDIR=/tmp
out=gpurun_out/r04b
mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.txt
timeout -k 10 300 python tools/fuzz_binned.py 0 150 6 > $out/fuzz_binned.txt 2>&1; echo "fuzz binned rc=$?"; tail -3 $out/fuzz_binned.txt
timeout -k 10 300 python tools/fuzz_sequence.py 0 100 > $out/fuzz_sequence.txt 2>&1; echo "fuzz sequence rc=$?"; tail -3 $out/fuzz_sequence.txt
timeout -k 10 300 python tools/band_cost.py 8 both > $out/band_cost.txt 2>&1; echo "band cost rc=$?"; cat $out/band_cost.txt
timeout -k 10 200 python tools/moving_light.py > $out/moving_light.txt 2>&1; echo "moving light rc=$?"; cat $out/moving_light.txt
timeout -k 10 300 python bench.py --workload soup100k --no-cpu-baseline > $out/bench_soup100k.json 2> $out/bench_soup100k.err; echo "bench rc=$?"; python3 -c "
import json;d=json.load(open('$out/bench_soup100k.json'));print({k:d[k] for k in ('value','ms_per_frame','kernel_ms_rank0','kernel_ms_alone_rank0','static_camera')})"
echo done
