#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc counter CSVs (one or more pass directories) as a markdown table.

    python tools/pmc_table.py gpurun_out/pmc_a gpurun_out/pmc_b [--match REGEX] > profiles/roundN_pmc_x.md
"""
import argparse
import collections
import csv
import glob
import re

ap = argparse.ArgumentParser()
ap.add_argument("dirs", nargs="+")
ap.add_argument("--match", default="")
a = ap.parse_args()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in a.dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if re.search(a.match, r["Kernel_Name"]):
                acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
# average duration of the same kernels from the kernel trace of the first pass directory (counters serialise kernels, not slow them)
for f in glob.glob(a.dirs[0] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if re.search(a.match, r["Kernel_Name"]) and r["Kernel_Name"][:70] in acc:
            acc[r["Kernel_Name"][:70]]["duration_us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
cols = sorted({c for v in acc.values() for c in v})
print("| kernel | launches | " + " | ".join(cols) + " |")
print("|---|---|" + "---|" * len(cols))
for k, v in sorted(acc.items()):
    n = max(len(x) for x in v.values())
    print(f"| `{k}` | {n} | " + " | ".join(f"{sum(v[c]) / len(v[c]):.4g}" if c in v else "-" for c in cols) + " |")
