#!/usr/bin/env bash
# tools/pmc_trace8k.sh <tag> -- (GPU box) what the kernels of the 1 M-triangle frame at 8K (BASELINE configs[4], whole frame on one GPU, one
# frame in flight, the view moving) wait for: four rocprofv3 --pmc passes (kernel-trace only, the program directly behind `--`), per
# launch, as JSON under gpurun_out/<tag>/pmc_summary.json.  L2 hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum); FETCH_SIZE is in
# KiB and counts 64 B per 128-B request on gfx950 (x2 for bytes); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles.
set -uo pipefail
export TMPDIR=/tmp
out="gpurun_out/${1:-trace8k}"
mkdir -p "$out"
run() { rocprofv3 --kernel-trace --pmc "${@:2}" --output-format csv -d "$out/$1" -- python3 tools/band_prof.py 0 4320 6 move > "$out/$1.txt" 2>&1; echo "$1 rc=$?"; }
run pmc_l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run pmc_sq SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run pmc_sq2 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS
run pmc_fetch FETCH_SIZE
python3 - "$out" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
for sub in ("pmc_l2", "pmc_sq", "pmc_sq2", "pmc_fetch"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(out + "/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:60]
            if "mirt" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        res.setdefault(k, {}).update({c: round(v / max(1, cnt[(k, c)]), 1) for c, v in d.items()})
        res[k]["launches_" + sub] = max(cnt[(k, c)] for c in d)
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1)
t = next((v for k, v in res.items() if "k_rt_trace2<false, false>" in k), None)
if t:
    print("k_rt_trace2: L2 hit rate %.2f, fetched %.2f GB per launch, waves parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES) %.2f, issue-active %.2f, VALU %.0f M SALU %.0f M per launch"
          % (t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"]), t["FETCH_SIZE"] * 2 * 1024 / 1e9, t["SQ_WAIT_ANY"] / t["SQ_WAVE_CYCLES"],
             t["SQ_ACTIVE_INST_ANY"] / t["SQ_WAVE_CYCLES"], t["SQ_INSTS_VALU"] / 1e6, t["SQ_INSTS_SALU"] / 1e6))
PY
find "$out" -name "*counter_collection.csv" -delete; find "$out" -name "*kernel_trace.csv" -delete; find "$out" -name "*agent_info.csv" -delete
