#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel from a counter_collection.csv under a directory: python tools/pmc_kernel.py DIR [name-filter]"""
import csv, glob, sys
from collections import defaultdict
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:90]
    if flt and flt not in k:
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, c in acc.items():
    print(k, "dispatches", len(n[k]))
    for name, v in sorted(c.items()):
        print(f"   {name:32s} {v / len(n[k]):16.0f} per dispatch")
