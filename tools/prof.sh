#!/usr/bin/env bash
# tools/prof.sh <tag> <bench args...>  -- rocprofv3 kernel-trace stats + one PMC pass of bench.py on the GPU box.
# Run through gpurun; outputs land in gpurun_out/prof_<tag>/ (copy the summaries worth keeping into profiles/).
set -uo pipefail
tag="$1"; shift
out="gpurun_out/prof_$tag"
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --no-cpu-baseline "$@" > "$out/bench_trace.json" 2> "$out/trace.err"
echo "trace rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$out/pmc" -- python3 bench.py --no-cpu-baseline "$@" > "$out/bench_pmc.json" 2> "$out/pmc.err"
echo "pmc rc=$?"
find "$out" -name "*kernel_stats.csv" | head -1 | xargs -r cat | head -30
