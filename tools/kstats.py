#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 --stats kernel_stats.csv found under a directory."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for r in list(csv.DictReader(open(f)))[:n]:
    print(f'{r["Name"][:100]:100s} {r["Calls"]:>6s} {float(r["TotalDurationNs"])/1e6:9.3f} ms  avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Percentage"]:>6s}%')
