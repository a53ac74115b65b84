#!/usr/bin/env python3
"""Per-kernel MFMA utilisation and HBM traffic of one bench.py step from four rocprofv3 passes (same command each time,
`python3 bench.py --steps 1 --warmup 1 --cpu-outfits 0`; counters in their own passes, MI355X_MICROARCH.md "rocprofv3 PMC slots"):

    rocprofv3 --kernel-trace --stats -d P/trace  --output-format csv -- python3 bench.py ...        durations
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES -d P/sq ...
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d P/fetch ...      rocprofv3 --kernel-trace --pmc WRITE_SIZE -d P/write ...

    python tools/pmc_mfma_util.py P > profiles/r02_mfma_util.json

Per kernel (all its dispatches summed):
  mfma_util      = sum SQ_VALU_MFMA_BUSY_CYCLES / (sum GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)   (rocprofv3's own MfmaUtil expression)
  mfma_tflops    = sum (MOPS_F16 + MOPS_BF16) * 512 FLOP / the kernel's time in the un-perturbed trace pass (EXECUTED FLOPs: split
                   weights and three-product GEMMs count every product)
  hbm_gb_s       = (2 * FETCH_SIZE + WRITE_SIZE) KiB / trace-pass time; FETCH_SIZE doubled for gfx950 (128-B requests tallied at 64 B,
                   MI355X_MICROARCH.md HBM section); Infinity-Cache hits are included in the fabric counters
"""
import csv, glob, json, os, re, sys
from collections import defaultdict

P = sys.argv[1]


def short(name):
    if name.startswith("_Z"):            # the stats csv carries mangled names for some kernels: <len>gemm_w2_kernel...
        m = re.search(r"\d+((?:gemm|fused|attention|set_attention|layernorm|splitk|pack_rows|patchify|vit_embed|fold_pack|stats_finalize|row_stats|gather|l2norm|text_|cp_head|multi_copy|iota|resample|fitb|topk|dist_tile)[A-Za-z0-9_]*?)(I|E|$)", name)
        if m:
            tail = name[m.end(1):]
            dt = "<f16>" if "DF16_" in tail else "<bf16>" if "DF16b" in tail else ""
            tpl = "".join(f",{v}" for v in re.findall(r"Li(\d+)E", tail))
            return m.group(1) + (dt[:-1] + tpl + ">" if dt else "")
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").strip()


def counters(d):
    fs = sorted(glob.glob(os.path.join(P, d, "**", "*counter_collection.csv"), recursive=True))
    out = defaultdict(lambda: defaultdict(float)); calls = defaultdict(set)
    if not fs:
        return out, calls
    for r in csv.DictReader(open(fs[-1])):
        k = short(r["Kernel_Name"])
        out[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
    return out, calls


stats = {}
fs = sorted(glob.glob(os.path.join(P, "trace", "**", "*kernel_stats.csv"), recursive=True))
for r in csv.DictReader(open(fs[-1])):
    k = short(r["Name"])
    e = stats.setdefault(k, {"calls": 0, "total_ns": 0.0})
    e["calls"] += int(r["Calls"]); e["total_ns"] += float(r["TotalDurationNs"])
sq, sq_calls = counters("sq")
fe, _ = counters("fetch")
wr, _ = counters("write")
sq2, _ = counters("sq2")          # optional pass: where the waves' cycles go
# algorithmic bytes per launch by kernel, from the un-profiled bench line of the same build (roofline.per_kernel)
alg = {}
try:
    bl = json.load(open(os.path.join(P, "bench_line.json")))
    alg = {k: v["algorithmic_mb_per_launch"] * 1e6 for k, v in bl["roofline"]["per_kernel"].items()}
except Exception:
    bl = None
rows = []
tot = {"ns": 0.0, "busy": 0.0, "gui": 0.0, "mops": 0.0, "bytes": 0.0}
for k, s in sorted(stats.items(), key=lambda kv: -kv[1]["total_ns"]):
    c = sq.get(k, {})
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    mops = (c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) + c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)) * 512.0
    scale = s["calls"] / max(len(sq_calls.get(k, ())), 1)          # the passes run the same command: same dispatch count
    byts = (2.0 * fe.get(k, {}).get("FETCH_SIZE", 0.0) + wr.get(k, {}).get("WRITE_SIZE", 0.0)) * 1024.0
    t = s["total_ns"] * 1e-9
    row = {"kernel": k, "calls": s["calls"], "avg_us": round(s["total_ns"] / s["calls"] / 1e3, 1), "total_ms": round(s["total_ns"] / 1e6, 3),
           "mfma_util": round(busy / (gui * 1024.0), 4) if gui else None,
           "mfma_tflops_executed": round(mops * scale / t / 1e12, 1) if mops else 0.0,
           "hbm_bytes_per_call": round(byts / max(s["calls"], 1)), "hbm_gb_s": round(byts / t / 1e9, 1) if byts else 0.0}
    def same_kernel(bench_name, prof_name):          # bench: gemm_big_kernel<2,2,1>, gemm_w2f8_kernel; profiler: gemm_big_kernel<f16,2,2,1,0>, gemm_w2f8_kernel<0>
        bb, _, ba = bench_name.partition("<"); pb, _, pa = prof_name.partition("<")
        if bb.strip() != pb.strip():
            return False
        pa = re.sub(r"^(f16|bf16),?", "", pa)
        return not ba or pa.startswith(ba.rstrip(">"))
    for name, b in alg.items():
        if same_kernel(name, k):
            # several profiler variants (operand type, tile height) share one bench row: the per-launch figure is the row's average
            row["algorithmic_bytes_per_call"] = round(b); row["traffic_over_algorithmic"] = round(row["hbm_bytes_per_call"] / b, 3) if b else None
    c2 = sq2.get(k, {})
    if c2.get("SQ_WAVE_CYCLES"):
        wc = c2["SQ_WAVE_CYCLES"]
        row["wave_cycle_shares"] = {"wait_any": round(c2.get("SQ_WAIT_ANY", 0.0) / wc, 3), "wait_inst_any": round(c2.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3),
                                    "active_inst_any": round(c2.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3)}
        row["lds_bank_conflict_over_lds_active"] = round(c2.get("SQ_LDS_BANK_CONFLICT", 0.0) / c2["SQ_LDS_IDX_ACTIVE"], 4) if c2.get("SQ_LDS_IDX_ACTIVE") else None
        if "SQ_INSTS_VALU_MFMA_MOPS_F8" in c2:
            row["mfma_mops_f8"] = c2["SQ_INSTS_VALU_MFMA_MOPS_F8"]
    rows.append(row)
    tot["ns"] += s["total_ns"]; tot["busy"] += busy; tot["gui"] += gui; tot["mops"] += mops * scale; tot["bytes"] += byts
gemm = [r for r in rows if "gemm" in r["kernel"] or "fused_qkv" in r["kernel"]]
g_ns = sum(r["total_ms"] for r in gemm) * 1e6
g_calls = sum(r["calls"] for r in gemm)
g_bytes = sum(r["hbm_bytes_per_call"] * r["calls"] for r in gemm)
print(json.dumps({
    "source": "rocprofv3 passes of `python3 bench.py --steps 1 --warmup 1 --cpu-outfits 0` (3 forward steps incl. the breakdown step): --kernel-trace --stats; "
              "--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES; --pmc FETCH_SIZE; --pmc WRITE_SIZE",
    "definitions": {"mfma_util": "sum SQ_VALU_MFMA_BUSY_CYCLES / (sum GRBM_GUI_ACTIVE / 8 * 1024 SIMDs), per kernel over all its dispatches",
                    "mfma_tflops_executed": "(MOPS_F16 + MOPS_BF16) * 512 / trace-pass time: every MFMA product counted (peak 2,500 dense)",
                    "hbm": "2 x FETCH_SIZE + WRITE_SIZE (KiB), gfx950 correction per MI355X_MICROARCH.md; per call and as GB/s over the trace-pass time (peak 8,000)"},
    "whole_run": {"kernel_time_ms": round(tot["ns"] / 1e6, 3), "mfma_util": round(tot["busy"] / (tot["gui"] * 1024.0), 4) if tot["gui"] else None,
                  "mfma_tflops_executed": round(tot["mops"] / (tot["ns"] * 1e-9) / 1e12, 1), "hbm_gb_s": round(tot["bytes"] / (tot["ns"] * 1e-9) / 1e9, 1)},
    "gemm_kernels": {"launches": g_calls, "avg_launch_us": round(g_ns / 1e3 / max(g_calls, 1), 2), "gemm_hbm_bytes_per_launch": round(g_bytes / max(g_calls, 1)),
                     "hbm_gb_s": round(g_bytes / (g_ns * 1e-9) / 1e9, 1) if g_ns else 0.0},
    "per_kernel_traffic": {r["kernel"]: {"hbm_bytes_per_call": r["hbm_bytes_per_call"], "algorithmic_bytes_per_call": r.get("algorithmic_bytes_per_call"),
                                         "traffic_over_algorithmic": r.get("traffic_over_algorithmic")} for r in gemm},
    "bench_line_of_the_same_build": ({k: bl[k] for k in ("value", "ms_per_step")} if bl else None),
    "kernels": rows}, indent=1))
