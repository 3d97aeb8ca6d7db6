"""Summarise rocprofv3 --pmc counter_collection.csv files per sweep kernel: dispatches, mean and SUM per counter (the pass families launch
many slices per half-step: their per-half-step figure is the sum over the run / the half-steps of the run).
usage: python tools/pmc_summary.py DIR [kernel-name regex] > profiles/rNN_xxx_pmc.md      (DIR is searched recursively)"""
import collections
import csv
import glob
import os
import re
import sys

root = sys.argv[1]
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else "sweep|tiled|pass|cached|dense|multi")
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(path)):
        if pat.search(r["Kernel_Name"]):
            rows[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("| kernel | counter | dispatches | mean | sum | last dispatch |\n|---|---|---|---|---|---|")
for k in sorted(rows):
    for c in sorted(rows[k]):
        v = rows[k][c]
        print(f"| `{k}` | {c} | {len(v)} | {sum(v) / len(v):.6g} | {sum(v):.6g} | {v[-1]:.6g} |")
