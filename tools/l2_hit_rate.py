#!/usr/bin/env python3
"""Per-kernel L2 hit rate from a rocprofv3 counter pass:  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d DIR -- <cmd>
    python tools/l2_hit_rate.py DIR      ->  kernel, dispatches, TCC_HIT_sum, TCC_MISS_sum, hit rate (MI355X_MICROARCH.md, L2 section)"""
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").replace("(anonymous namespace)::", "").strip()[:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, c in sorted(acc.items(), key=lambda kv: -(kv[1].get("TCC_HIT_sum", 0) + kv[1].get("TCC_MISS_sum", 0))):
    h, m = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
    if h + m > 0:
        print(f"{k:62s} dispatches {len(n[k]):5d}  hit {h:.3e}  miss {m:.3e}  hit rate {h / (h + m):.3f}  requests per dispatch {(h + m) / len(n[k]):.3e}")
