#!/usr/bin/env python3
"""Post-process two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; --output-format csv) of `bench.py --steps 1 --warmup 1`
into HBM bytes per GEMM launch -> profiles/r01_traffic_pmc.json.  Counters are in KiB; FETCH_SIZE is doubled for gfx950
(MI355X_MICROARCH.md, HBM section: 128-B requests are tallied at 64 B).

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write [steps_profiled]
"""
import csv, glob, json, sys

def load(d, counter):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
    tot_all = tot_gemm = 0.0; n_gemm = 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        v = float(r["Counter_Value"]) * 1024.0
        tot_all += v
        if "gemm_" in r["Kernel_Name"] and "splitk_reduce" not in r["Kernel_Name"]:
            tot_gemm += v; n_gemm += 1
    return tot_all, tot_gemm, n_gemm

fa, fg, n1 = load(sys.argv[1], "FETCH_SIZE")
wa, wg, n2 = load(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2      # warmup 1 + steps 1 (+ the untimed breakdown step is excluded by --no-breakdown)
assert n1 == n2, (n1, n2)
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 1 --warmup 1`, kernel-trace csv; all profiled steps averaged",
       "unit_note": "counters in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section (gfx950 tallies 128-B requests at 64 B)",
       "gemm_launches_profiled": n1,
       "gemm_fetch_bytes_per_launch": 2 * fg / n1, "gemm_write_bytes_per_launch": wg / n1,
       "gemm_hbm_bytes_per_launch": (2 * fg + wg) / n1,
       "all_kernels_fetch_bytes": 2 * fa, "all_kernels_write_bytes": wa}
print(json.dumps(out, indent=1))
