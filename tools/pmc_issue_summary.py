"""tools/pmc_issue_summary.py <gpurun_out/pmcv_TAG> -- per-kernel averages of the issue-side counters of tools/pmc_valu.sh
(two rocprofv3 --pmc passes), as JSON {kernel: {counter: average per launch}}."""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for sub in ("pmc1", "pmc2"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (root, sub), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:80]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
print(json.dumps({k: dict({c: round(v / max(1, cnt[(k, c)]), 1) for c, v in d.items()}, launches=max(cnt[(k, c)] for c in d)) for k, d in acc.items() if "mirt" in k}, indent=1))
