"""tools/store_profiles.py <round tag, e.g. r02> -- copy what tools/collect_profiles.sh left in gpurun_out/ into profiles/."""
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
# (the digest of the device code each summary was collected at lies BESIDE it, written on the GPU box by tools/collect_profiles.sh:
# it is copied, never recomputed here -- the tree may have moved on since)
for f in glob.glob("gpurun_out/bench_*.json"):
    if os.path.getsize(f) > 0:
        shutil.copy(f, "profiles/%s_%s" % (tag, os.path.basename(f)))
traffic = {"_comment": "HBM bytes per launch from rocprofv3 PMC passes (tools/pmc_hbm.sh: FETCH_SIZE and WRITE_SIZE in separate passes, "
                       "KiB units, FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM); summarised by tools/pmc_summary.py (every "
                       "kernel of the run). MI355X, bench.py defaults (moving camera, frames in flight).",
           "stamps": {}, "workloads": {}}
issue = {"_comment": "issue-side PMC counters per launch (tools/pmc_valu.sh: two rocprofv3 --pmc passes, kernel-trace only), averaged "
                     "over the launches by tools/pmc_issue_summary.py. MI355X, bench.py defaults.",
         "stamps": {}, "workloads": {}}
# a partial collection (tools/collect_profiles.sh pmc <workload> ...) adds to what the round has already stored
for doc, name in ((traffic, "hbm_traffic"), (issue, "pmc_issue")):
    path = "profiles/%s_%s.json" % (tag, name)
    if os.path.exists(path):
        old = json.load(open(path))
        doc["workloads"].update(old.get("workloads", {}))
        doc["stamps"].update(old.get("stamps", {}))
for t in ("cornell1080", "soup100k", "raster4k", "soup1m8k", "raster4kdof8", "cornell1080dof8", "cornell1080soft16", "cornell1080aa3", "cornell500"):
    stats = sorted(glob.glob("gpurun_out/prof_%s/trace/*/*_kernel_stats.csv" % t), key=os.path.getmtime)
    if stats:
        shutil.copy(stats[-1], "profiles/%s_rocprof_%s_kernel_stats.csv" % (tag, t))
    for src, doc in (("gpurun_out/pmc_%s/summary.json" % t, traffic), ("gpurun_out/pmcv_%s/summary.json" % t, issue)):
        stamp = os.path.join(os.path.dirname(src), "csrc_sha16.txt")
        if os.path.exists(src) and os.path.getsize(src) > 2 and os.path.exists(stamp):
            doc["workloads"][t] = json.load(open(src))
            doc["stamps"][t] = open(stamp).read().strip()
        elif os.path.exists(src):
            print("skipping %s: no digest beside it (collected by an older tools/collect_profiles.sh?)" % src)
if traffic["workloads"]:
    json.dump(traffic, open("profiles/%s_hbm_traffic.json" % tag, "w"), indent=1)
if issue["workloads"]:
    json.dump(issue, open("profiles/%s_pmc_issue.json" % tag, "w"), indent=1)
for name in ("ubench", "edgebench", "divcheck", "moving_light", "fuzz_binned_vs_brute", "fuzz_small_scenes_vs_oracle", "fuzz_call_sequences", "fuzz_raster_call_sequences"):
    src = "gpurun_out/%s.txt" % name
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copy(src, "profiles/%s_%s.txt" % (tag, name))
print(sorted(f for f in os.listdir("profiles") if f.startswith(tag)))
