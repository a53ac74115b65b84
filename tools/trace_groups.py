#!/usr/bin/env python3
"""Group a rocprofv3 kernel-trace csv by (kernel, grid size): calls, mean / min duration.  python tools/trace_groups.py trace.csv"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
g = collections.defaultdict(list)
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "")[:60]
    grid = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])))
    g[(name, grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in g.values())
for (name, grid), v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    print(f"{name:62s} grid {grid[0]:5d}x{grid[1]:<3d} calls {len(v):5d}  mean {sum(v)/len(v)/1e3:7.2f} us  min {min(v)/1e3:7.2f}  total {sum(v)/1e6:8.3f} ms ({100*sum(v)/tot:4.1f} %)")
