#!/usr/bin/env bash
# tools/pmc_hbm.sh <tag> <bench args...> -- HBM traffic counters of every kernel of a bench.py run, collected as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (they do not fit one
# pass: 3 + 2 of the 4 TCC slots), kernel-trace only, no other trace domain.  Units are KiB; on gfx950 FETCH_SIZE
# counts 64 B per 128-B request of wide coalesced reads, so tools/pmc_summary.py doubles it (the guide's correction).
set -uo pipefail
tag="$1"; shift
out="gpurun_out/pmc_$tag"
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 bench.py --no-cpu-baseline "$@" > "$out/bench_fetch.json" 2> "$out/fetch.err"
echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 bench.py --no-cpu-baseline "$@" > "$out/bench_write.json" 2> "$out/write.err"
echo "write rc=$?"
