#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSV output: mean counter value per kernel name (per dispatch)."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "ofarn"
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pat not in k:
            continue
        short = k.split("(")[0].replace("ofarn::", "")
        short += "@" + r.get("Grid_Size", "")
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"    {c:28s} n={len(v):4d} mean={sum(v) / len(v):16.1f}")
