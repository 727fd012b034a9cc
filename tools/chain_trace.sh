#!/usr/bin/env bash
# tools/chain_trace.sh [soup100k|soup1m8k] -- (GPU box) the kernels of ONE frame in flight, camera moving, as a timeline: rocprofv3 --kernel-trace
# over tools/moving_ab.py's child with one frame in flight, then tools/timeline.py over the last frames (start, duration, gap to the previous kernel).
set -uo pipefail
export TMPDIR=/tmp
work="${1:-soup100k}"
out=gpurun_out/chain_$work
rm -rf "$out"
MIRT_AB_FLIGHT=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$out" -- python3 tools/moving_ab.py --child "$work" - > "$out.txt" 2>&1
f=$(find "$out" -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py "$f" 3
