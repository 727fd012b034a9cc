#!/usr/bin/env bash
# tools/fuzz_campaign.sh <first seed> -- (GPU box) a second, larger set of seeds through the four fuzzers on the round's final code; the
# summary lines go to gpurun_out/fuzz_more_seeds.txt (kept as profiles/<round>_fuzz_more_seeds.txt).
set -uo pipefail
s="${1:-20000}"
out=gpurun_out/fuzz_more_seeds.txt
: > $out
python tools/fuzz_binned.py $s 500 6 > gpurun_out/fm_binned.txt 2>&1; echo "fuzz_binned rc=$?"; tail -1 gpurun_out/fm_binned.txt >> $out
python tools/fuzz_small.py $s 6000 > gpurun_out/fm_small.txt 2>&1; echo "fuzz_small rc=$?"; tail -1 gpurun_out/fm_small.txt >> $out
python tools/fuzz_sequence.py $s 300 > gpurun_out/fm_seq.txt 2>&1; echo "fuzz_sequence rc=$?"; tail -1 gpurun_out/fm_seq.txt >> $out
python tools/fuzz_raster_sequence.py $s 300 > gpurun_out/fm_rseq.txt 2>&1; echo "fuzz_raster_sequence rc=$?"; tail -1 gpurun_out/fm_rseq.txt >> $out
grep -h MISMATCH gpurun_out/fm_*.txt | head -20 >> $out
cat $out
