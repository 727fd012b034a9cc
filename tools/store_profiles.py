"""tools/store_profiles.py <round tag, e.g. r01> -- copy what tools/collect_profiles.sh left in gpurun_out/ into profiles/."""
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
for f in glob.glob("gpurun_out/bench_*.json"):
    if os.path.getsize(f) > 0:
        shutil.copy(f, "profiles/%s_%s" % (tag, os.path.basename(f)))
traffic = {"_comment": "HBM bytes per launch from rocprofv3 PMC passes (tools/pmc_hbm.sh: FETCH_SIZE and WRITE_SIZE in separate passes, "
                       "KiB units, FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM); summarised by tools/pmc_summary.py. MI355X.",
           "workloads": {}}
for t in ("cornell1080", "soup100k", "raster4k"):
    stats = sorted(glob.glob("gpurun_out/prof_%s/trace/*/*_kernel_stats.csv" % t), key=os.path.getmtime)
    if stats:
        shutil.copy(stats[-1], "profiles/%s_rocprof_%s_kernel_stats.csv" % (tag, t))
    s = "gpurun_out/pmc_%s/summary.json" % t
    if os.path.exists(s) and os.path.getsize(s) > 2:
        traffic["workloads"][t] = json.load(open(s))
if traffic["workloads"]:
    json.dump(traffic, open("profiles/%s_hbm_traffic.json" % tag, "w"), indent=1)
for t in ("cornell1080", "raster4kdof8"):
    src = "gpurun_out/pmcv_%s.txt" % t
    if os.path.exists(src) and os.path.getsize(src) > 0:
        # keep our kernels' lines only (kernel-trace stats + the two issue-side PMC passes)
        keep = [l for l in open(src) if ("mirt::" in l or l.startswith('"Name"') or l.startswith("trace rc") or l.startswith("pmc"))]
        open("profiles/%s_pmc_issue_%s.txt" % (tag, t), "w").writelines(keep)
if os.path.exists("gpurun_out/ubench.txt") and os.path.getsize("gpurun_out/ubench.txt") > 0:
    shutil.copy("gpurun_out/ubench.txt", "profiles/%s_ubench_valu_lds.txt" % tag)
if os.path.exists("gpurun_out/sortbench.txt") and os.path.getsize("gpurun_out/sortbench.txt") > 0:
    shutil.copy("gpurun_out/sortbench.txt", "profiles/%s_sortbench.txt" % tag)
print(sorted(os.listdir("profiles")))
