#!/usr/bin/env bash
# tools/raster_chain.sh [libmirt variant.so] -- (GPU box) the kernels of the rasterised 4K Cornell frame, one frame in flight, as a timeline
# (rocprofv3 --kernel-trace over tools/raster_prof.py, then tools/timeline.py over the last frames).
set -uo pipefail
export TMPDIR=/tmp
out=gpurun_out/rchain_${2:-x}
rm -rf "$out"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$out" -- python3 tools/raster_prof.py ${1:-} > "$out.txt" 2>&1
f=$(find "$out" -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py "$f" 2
