#!/usr/bin/env bash
# tools/rehearse_ranks.sh -- (one-GPU box) bench.py's N > 1 control flow with every rank on device 0 (MIRT_BENCH_REHEARSAL=1: gloo group,
# host-staged gathers, assembled frames checked against a single-GPU frame).  Not a measurement: what it shows is that the sharded
# frames of 2 and 3 ranks come out identical to the one-GPU frame on the code as it stands.
set -uo pipefail
export MIRT_BENCH_REHEARSAL=1 MIRT_BENCH_TARGET_S=0.2
out=gpurun_out/rehearsal.txt
: > $out
for spec in "2 soup100k" "3 soup100k" "2 cornell1080" "3 raster4k" "3 soup1m8k"; do
  set -- $spec
  timeout -k 10 240 python bench.py --gpus $1 --workload $2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/rehearsal_$1_$2.json 2> gpurun_out/rehearsal_$1_$2.err
  rc=$?
  echo "ranks $1 workload $2 rc=$rc $(python - gpurun_out/rehearsal_$1_$2.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value %.1f %s, %s" % (d["value"], d["unit"], {k: d[k] for k in d if "check" in k or "rehears" in k or k == "n_gpus"}))
except Exception as e:
    print("no line:", e)
PY
)" | tee -a $out
  [ $rc -eq 0 ] || { tail -5 gpurun_out/rehearsal_$1_$2.err | tee -a $out; break; }
done
