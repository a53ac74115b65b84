#!/usr/bin/env python3
"""Timeline of the headline step from a rocprofv3 kernel trace of `bench.py` (default command: text tower on the side stream).
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 4 --warmup 2 --cpu-outfits 0
    python tools/trace_timeline.py $(find DIR -name "*kernel_trace.csv")
Steps are cut at the ViT's patchify kernel.  Per step: wall span, the busy time and the idle gaps of the queue that runs the ViT, what
runs after the ViT's last kernel (the set transformer and the heads: nothing can overlap them inside one step), and how much of the other
queue's kernel time falls inside the ViT's span."""
import csv, sys, collections, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
qkey = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
marks = [i for i, r in enumerate(rows) if "patchify" in r["Kernel_Name"]]
print(f"{len(rows)} kernels, {len(marks)} steps (cut at patchify); queue column: {qkey}")
short = lambda n: n.split("(")[0].replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")[:40]
for a, b in zip(marks, marks[1:] + [len(rows)]):
    seg = rows[a:b]
    if b == len(rows):                                   # last step: drop what follows the CP head (parity legs, breakdown step)
        ends = [i for i, r in enumerate(seg) if "cp_head" in r["Kernel_Name"]]
        seg = seg[:ends[0] + 1] if ends else seg
    span = (max(r["e"] for r in seg) - seg[0]["s"]) / 1e6
    vq = seg[0][qkey] if qkey else None
    vit = [r for r in seg if (not qkey) or r[qkey] == vq]
    oth = [r for r in seg if qkey and r[qkey] != vq]
    last_w2 = max((i for i, r in enumerate(vit) if "gemm_w2f8" in r["Kernel_Name"]), default=len(vit) - 1)
    tower_end = vit[last_w2]["e"]
    gaps = [(vit[i + 1]["s"] - vit[i]["e"]) / 1e3 for i in range(len(vit) - 1)]
    pos = [g for g in gaps if g > 0]
    big = sorted(((g, short(vit[i]["Kernel_Name"]), short(vit[i + 1]["Kernel_Name"])) for i, g in enumerate(gaps) if g > 20), reverse=True)[:6]
    busy = sum(r["e"] - r["s"] for r in vit) / 1e6
    w2 = sum(r["e"] - r["s"] for r in vit if "gemm_w2f8" in r["Kernel_Name"]) / 1e6
    tail = (max(r["e"] for r in seg) - tower_end) / 1e6
    ob = sum(r["e"] - r["s"] for r in oth) / 1e6
    oin = sum(max(0, min(r["e"], tower_end) - max(r["s"], seg[0]["s"])) for r in oth) / 1e6
    print(f"step: span {span:6.2f} ms | ViT queue: {len(vit)} kernels busy {busy:6.2f} (gemm_w2f8 {w2:6.2f}), idle gaps {sum(pos) / 1e3:5.2f} ms (median {statistics.median(gaps):.1f} us) | "
          f"after the last gemm_w2f8: {tail:5.2f} ms | other queue: {len(oth)} kernels, {ob:5.2f} ms of kernel time, {oin:5.2f} inside the ViT's span")
    for g, x, y in big:
        print(f"        gap {g:7.1f} us between {x} -> {y}")
