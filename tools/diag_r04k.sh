#!/usr/bin/env bash
set -uo pipefail
export TMPDIR=/tmp
out=gpurun_out/r04k
mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 $out/pytest.txt
timeout -k 10 300 python tools/fuzz_sequence.py 6000 150 > $out/fuzz_sequence.txt 2>&1; echo "fuzz sequence rc=$?"; tail -2 $out/fuzz_sequence.txt
timeout -k 10 300 python tools/fuzz_binned.py 6000 100 6 > $out/fuzz_binned.txt 2>&1; echo "fuzz binned rc=$?"; tail -2 $out/fuzz_binned.txt
timeout -k 10 200 python tools/moving_light.py > $out/moving_light.txt 2>&1; echo "moving light rc=$?"; cat $out/moving_light.txt
MIRT_LIGHT_SIDE_STREAM=0 timeout -k 10 200 python tools/moving_light.py > $out/moving_light_noside.txt 2>&1; echo "no side stream:"; cat $out/moving_light_noside.txt
echo done
