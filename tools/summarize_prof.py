#!/usr/bin/env python3
"""Summarise a tools/profile_bench.sh output dir: kernel stats + per-kernel mean PMC values.
Usage: tools/summarize_prof.py gpurun_out/prof_<tag> [--traffic-json profiles/traffic_latest.json] > profiles/<name>.md

HBM traffic follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are collected in separate
--pmc passes, are in KiB, and are reported raw; the guide's x2 correction of FETCH_SIZE applies to wide
16-B-per-lane streams only, this kernel reads bytes and dwords, so the raw figure is a LOWER bound on read
traffic and (2*FETCH+WRITE) an upper bound -- both are written out."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
tj = sys.argv[sys.argv.index("--traffic-json") + 1] if "--traffic-json" in sys.argv else None
print(f"# rocprofv3 summary of `{os.path.basename(d)}`\n")
def newest(pattern):
    """gpurun MERGES a run's files into the local directory, so earlier runs' files (other process ids in their names) may still lie
    there: per directory only the most recent file counts"""
    by_dir = {}
    for f in glob.glob(pattern):
        k = os.path.dirname(f)
        if k not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[k]):
            by_dir[k] = f
    return sorted(by_dir.values())


for f in newest(os.path.join(d, "trace", "*", "*kernel_stats.csv")):
    print("## kernel stats (rocprofv3 --kernel-trace --stats)\n")
    rows = list(csv.DictReader(open(f)))
    print("| kernel | calls | total ns | avg ns | min ns | max ns | % |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r['Name'][:80]} | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {float(r['Percentage']):.2f} |")
acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))  # kernel -> counter -> dispatch -> value
for f in newest(os.path.join(d, "pmc*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "bmh" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
print("\n## PMC counters (mean per dispatch, per kernel)\n")
traffic = {}
for k in sorted(acc):
    print(f"### {k}\n")
    print("| counter | mean per dispatch | dispatches |")
    print("|---|---|---|")
    means = {}
    for c in sorted(acc[k]):
        vals = list(acc[k][c].values())
        means[c] = sum(vals) / len(vals)
        print(f"| {c} | {means[c]:.6g} | {len(vals)} |")
    if "FETCH_SIZE" in means and "WRITE_SIZE" in means:
        lo = (means["FETCH_SIZE"] + means["WRITE_SIZE"]) * 1024
        hi = (2 * means["FETCH_SIZE"] + means["WRITE_SIZE"]) * 1024
        print(f"\nHBM bytes per launch: {lo:.4g} (raw FETCH+WRITE) .. {hi:.4g} (FETCH doubled)\n")
        traffic[k] = {"hbm_bytes_per_launch": lo, "hbm_bytes_per_launch_fetch_x2": hi,
                      "FETCH_SIZE_KiB": means["FETCH_SIZE"], "WRITE_SIZE_KiB": means["WRITE_SIZE"],
                      "valu_insts_per_launch": means.get("SQ_INSTS_VALU"), "salu_insts_per_launch": means.get("SQ_INSTS_SALU"),
                      "waves_per_launch": means.get("SQ_WAVES")}
    print()
if tj:
    commit = os.popen("git -C %s rev-parse --short HEAD 2>/dev/null" % os.path.dirname(os.path.abspath(__file__))).read().strip()
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from csrc_sha import csrc_sha
    json.dump({"source": os.path.basename(d), "commit": commit or "unknown (no git on the GPU box: see the profile's file name)",
               "csrc_sha": csrc_sha(), "kernels": traffic},
              open(tj, "w"), indent=1)
