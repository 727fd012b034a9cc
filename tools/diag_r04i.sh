#!/usr/bin/env bash
set -uo pipefail
export TMPDIR=/tmp
out=gpurun_out/r04i
mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 $out/pytest.txt
timeout -k 10 300 python tools/fuzz_binned.py 4000 200 6 > $out/fuzz_binned.txt 2>&1; echo "fuzz binned rc=$?"; tail -3 $out/fuzz_binned.txt
timeout -k 10 300 python tools/fuzz_sequence.py 4000 120 > $out/fuzz_sequence.txt 2>&1; echo "fuzz sequence rc=$?"; tail -3 $out/fuzz_sequence.txt
timeout -k 10 200 python tools/moving_light.py > $out/moving_light.txt 2>&1; echo "moving light rc=$?"; cat $out/moving_light.txt
MIRT_BENCH_TARGET_S=0.3 timeout -k 10 300 python bench.py --workload soup100k --no-cpu-baseline > $out/bench_soup100k.json 2> $out/err.txt; python3 -c "
import json;d=json.load(open('$out/bench_soup100k.json'));print({k:d[k] for k in ('value','ms_per_frame','kernel_ms_rank0','kernel_ms_alone_rank0','static_camera')})"
MIRT_BENCH_TARGET_S=0.3 timeout -k 10 300 python bench.py --workload raster4k --no-cpu-baseline > $out/bench_raster4k.json 2> $out/err.txt; python3 -c "
import json;d=json.load(open('$out/bench_raster4k.json'));print({k:d[k] for k in ('value','ms_per_frame','kernel_ms_rank0','kernel_ms_alone_rank0')})"
rocprofv3 --kernel-trace --output-format csv -d $out/tl_light -- python3 tools/moving_light.py light > $out/tl_light.txt 2>&1
python3 tools/timeline.py $(find $out/tl_light -name "*kernel_trace.csv" | head -1) 2
find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
echo done
