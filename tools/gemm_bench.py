#!/usr/bin/env python3
"""GEMM micro-benchmark on the path's real shapes: interleaved rounds of tuning variants in ONE
process (guide rule 24), random data (rule 25).  python tools/gemm_bench.py [knob=value,...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outfitx_amd import _lib as L

# (name, M, N, K, epilogue) — epilogue as the model uses it: "b" bias->bf16, "r" bias+fp32 residual in place, "g" bias+quick_gelu->bf16, "f" fp32 out
SHAPES = [("vit qkv", 102400, 2304, 768, "b"), ("vit out", 102400, 768, 768, "r"), ("vit fc1", 102400, 3072, 768, "g"),
          ("vit fc2", 102400, 768, 3072, "r"), ("patch", 100352, 768, 3072, "f"), ("txt qkv", 16384, 1536, 512, "b"),
          ("txt out", 16384, 512, 512, "r"), ("txt fc1", 16384, 2048, 512, "g"), ("txt fc2", 16384, 512, 2048, "r"),
          ("set qkv x3", 2304, 3072, 3072, "f"), ("set fc2 x3", 2304, 1024, 6144, "r"),
          # the three-product text tower (K' = 3K): qkv (fp32 out), out-proj, fc1, fc2
          ("t3 qkv", 16384, 1536, 1536, "f"), ("t3 out", 16384, 512, 1536, "r"), ("t3 fc1", 16384, 2048, 1536, "g"), ("t3 fc2", 16384, 512, 6144, "r")]


def main():
    lib = L.load()
    variants = [dict(kv.split("=") for kv in v.split(",")) for v in sys.argv[1:] if "=" in v] or [{"0": "1"}, {"0": "8"}]
    only = [v for v in sys.argv[1:] if "=" not in v]
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for name, M, N, K, ep in SHAPES:
        if only and not any(o in name for o in only):
            continue
        A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
        C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16 if ep in "bg" else torch.float32)
        C.normal_(generator=g) if ep == "r" else None
        bias = torch.randn(N, device="cuda", generator=g)
        resid = C.data_ptr() if ep == "r" else None
        act, okind = (1 if ep == "g" else 0), (1 if ep in "bg" else 0)
        res = {i: [] for i in range(len(variants))}
        for rnd in range(6):
            for i, v in enumerate(variants):
                for k, val in v.items():
                    lib.ofx_tune(int(k), int(val))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 5
                e0.record()
                for _ in range(reps):
                    L.check(lib.ofx_gemm(A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), resid, M, N, K, K, N, N, act, okind, 1, s))
                e1.record(); e1.synchronize()
                if rnd:
                    res[i].append(e0.elapsed_time(e1) / reps)
        line = f"{name:12s} M={M:6d} N={N:5d} K={K:5d} ep={ep} "
        for i, v in enumerate(variants):
            t = np.median(res[i]); line += f"| {v} {t*1e3:8.1f} us {2*M*N*K/t/1e9:7.1f} TF "
        print(line, flush=True)


if __name__ == "__main__":
    main()
