#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (kernel stats CSV, PMC counter CSVs) into one small text summary for
profiles/.  Usage: summarize_profile.py OUT.txt --stats DIR [--pmc DIR ...] [--note TEXT]"""
import argparse
import collections
import csv
import glob
import os

ap = argparse.ArgumentParser()
ap.add_argument("out")
ap.add_argument("--stats")
ap.add_argument("--pmc", action="append", default=[])
ap.add_argument("--note", default="")
ap.add_argument("--filter", default="bwd_,lin_,forward_kernel,cost_kernel,select_kernel,rollout_kernel,eq_")
a = ap.parse_args()
keys = [k for k in a.filter.split(",") if k]
lines = []
if a.note:
    lines += [a.note, ""]
if a.stats:
    for f in glob.glob(os.path.join(a.stats, "**", "*kernel_stats.csv"), recursive=True):
        lines.append(f"# rocprofv3 --kernel-trace --stats : {os.path.relpath(f)}")
        lines.append(f"{'kernel':60s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
        for row in csv.DictReader(open(f)):
            name = row["Name"]
            short = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:60]
            lines.append(f"{short:60s} {row['Calls']:>7s} {float(row['TotalDurationNs']) / 1e6:10.3f} {float(row['AverageNs']) / 1e3:10.2f} "
                         f"{float(row['MinNs']) / 1e3:9.2f} {float(row['MaxNs']) / 1e3:9.2f} {float(row['Percentage']):6.2f}")
        lines.append("")
for d in a.pmc:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:60]
            if not any(s in k for s in keys):
                continue
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[(k, row["Counter_Name"])] += 1
        lines.append(f"# rocprofv3 --pmc : {os.path.relpath(f)}  (mean per dispatch)")
        for k, v in agg.items():
            for c, val in v.items():
                mean = val / cnt[(k, c)]
                extra = ""
                if c == "FETCH_SIZE":
                    extra = f"  -> {mean * 1024 / 1e6:.1f} MB as counted; x2 (gfx950 wide-read correction) = {2 * mean * 1024 / 1e6:.1f} MB"
                if c == "WRITE_SIZE":
                    extra = f"  -> {mean * 1024 / 1e6:.1f} MB"
                lines.append(f"{k:60s} {c:24s} {mean:16.1f} over {cnt[(k, c)]} dispatches{extra}")
        lines.append("")
open(a.out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:60]))
