#!/usr/bin/env python3
"""Where do the slow steps of a bench run come from?  From a rocprofv3 kernel trace: (a) device-wide idle gaps (no kernel of any queue running)
longer than 2 ms, with the kernels on either side; (b) kernels that took more than 3x their name's median.
    python tools/trace_stalls.py KERNEL_TRACE.csv"""
import csv, sys, collections, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
short = lambda n: n.split("(")[0].replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")[:44]
t0 = rows[0]["s"]
print(f"{len(rows)} kernels over {(rows[-1]['e'] - t0) / 1e6:.1f} ms")
end = rows[0]["e"]; last = rows[0]
gaps = []
for r in rows[1:]:
    if r["s"] > end + 2_000_000:
        gaps.append(((r["s"] - end) / 1e6, (end - t0) / 1e6, short(last["Kernel_Name"]), short(r["Kernel_Name"])))
    if r["e"] > end:
        end, last = r["e"], r
print(f"device-wide idle gaps > 2 ms: {len(gaps)}, {sum(g[0] for g in gaps):.1f} ms in all")
for g in gaps[:40]:
    print(f"   {g[0]:8.2f} ms idle at t = {g[1]:9.1f} ms   after {g[2]}  before {g[3]}")
by = collections.defaultdict(list)
for r in rows:
    by[short(r["Kernel_Name"])].append(r["e"] - r["s"])
med = {k: statistics.median(v) for k, v in by.items()}
slow = [(r["e"] - r["s"], r) for r in rows if (r["e"] - r["s"]) > 3 * med[short(r["Kernel_Name"])] and (r["e"] - r["s"]) > 1_000_000]
print(f"kernels over 3x their median and over 1 ms: {len(slow)}, {sum(d for d, _ in slow) / 1e6:.1f} ms in all")
for d, r in sorted(slow, key=lambda x: -x[0])[:30]:
    print(f"   {d / 1e6:8.2f} ms (median {med[short(r['Kernel_Name'])] / 1e6:.3f})  {short(r['Kernel_Name'])}  at t = {(r['s'] - t0) / 1e6:9.1f} ms")
