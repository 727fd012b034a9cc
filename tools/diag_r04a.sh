#!/usr/bin/env bash
# round 4, first look: per-kernel time of one band of config 5 (what a rank of the 8-way split runs) and what k_rt_trace2 waits for at 8K
set -uo pipefail
export TMPDIR=/tmp
out=gpurun_out/r04a
mkdir -p $out
for band in "0 540" "1620 2160" "0 4320"; do
  tag=$(echo $band | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/band_$tag -- python3 tools/band_prof.py $band 12 move > $out/band_$tag.txt 2>&1; echo "band $band rc=$?"
  find $out/band_$tag -name "*kernel_stats.csv" | head -1 | xargs -r cat > $out/band_${tag}_kernel_stats.csv
  find $out/band_$tag -name "*kernel_trace.csv" -delete
done
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $out/pmc_l2 -- python3 tools/band_prof.py 0 4320 6 move > $out/pmc_l2.txt 2>&1; echo "pmc l2 rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_sq -- python3 tools/band_prof.py 0 4320 6 move > $out/pmc_sq.txt 2>&1; echo "pmc sq rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $out/pmc_sq2 -- python3 tools/band_prof.py 0 4320 6 move > $out/pmc_sq2.txt 2>&1; echo "pmc sq2 rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 tools/band_prof.py 0 4320 6 move > $out/pmc_fetch.txt 2>&1; echo "pmc fetch rc=$?"
python3 - $out <<'PY'
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
print(json.dumps({k: v for k, v in res.items() if "trace2" in k or "bin_pairs" in k}, indent=1))
PY
find $out -name "*counter_collection.csv" -delete
find $out -name "*kernel_trace.csv" -delete
find $out -name "*agent_info.csv" -delete
echo done
