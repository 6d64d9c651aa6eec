#!/usr/bin/env python3
"""Summarise a tools/profile_bench.sh output dir: kernel stats + per-dispatch mean PMC values
for our kernels.  Usage: tools/summarize_prof.py gpurun_out/prof_<tag> [> profiles/<name>.md]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
print(f"# rocprofv3 summary of `{os.path.basename(d)}`\n")
for f in glob.glob(os.path.join(d, "trace", "*", "*kernel_stats.csv")):
    print("## kernel stats (rocprofv3 --kernel-trace --stats)\n")
    rows = list(csv.DictReader(open(f)))
    print("| kernel | calls | total ns | avg ns | min ns | max ns | % |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r['Name'][:70]} | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {float(r['Percentage']):.2f} |")
print("\n## PMC counters (mean per dispatch of kernels matching 'bmh')\n")
print("| counter | mean per dispatch | dispatches |")
print("|---|---|---|")
for f in sorted(glob.glob(os.path.join(d, "pmc*", "*", "*counter_collection.csv"))):
    acc = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "bmh" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for c, per in acc.items():
        vals = list(per.values())
        print(f"| {c} | {sum(vals)/len(vals):.6g} | {len(vals)} |")
