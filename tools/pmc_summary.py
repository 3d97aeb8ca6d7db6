"""Summarise rocprofv3 --pmc counter_collection.csv files (one directory per pass) per sweep kernel.
usage: python tools/pmc_summary.py gpurun_out > profiles/rNN_pmc_summary.md"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
rows = collections.defaultdict(dict)
for path in sorted(glob.glob(os.path.join(root, "pmc_*", "pmc_counter_collection.csv"))):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if "sweep" in r["Kernel_Name"] or "tiled" in r["Kernel_Name"]:
            per[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in per.items():
        for c, v in cs.items():
            rows[k][c] = (len(v), sum(v) / len(v), v[-1])
print("| kernel | counter | dispatches | mean | last dispatch |\n|---|---|---|---|---|")
for k in sorted(rows):
    for c in sorted(rows[k]):
        n, mean, last = rows[k][c]
        print(f"| `{k}` | {c} | {n} | {mean:.6g} | {last:.6g} |")
