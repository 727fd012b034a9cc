#!/usr/bin/env bash
set -uo pipefail
export TMPDIR=/tmp
out=gpurun_out/r04e
mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -8 $out/pytest.txt
rocprofv3 --kernel-trace --output-format csv -d $out/tl_band -- python3 tools/band_prof.py 0 540 6 move > $out/tl_band.txt 2>&1
python3 tools/timeline.py $(find $out/tl_band -name "*kernel_trace.csv" | head -1) 2
rocprofv3 --kernel-trace --output-format csv -d $out/tl_100k -- python3 tools/trace_prof.py move > $out/tl_100k.txt 2>&1
python3 tools/timeline.py $(find $out/tl_100k -name "*kernel_trace.csv" | head -1) 2
rocprofv3 --kernel-trace --output-format csv -d $out/tl_light -- python3 tools/moving_light.py light > $out/tl_light.txt 2>&1
python3 tools/timeline.py $(find $out/tl_light -name "*kernel_trace.csv" | head -1) 2
find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
echo done
