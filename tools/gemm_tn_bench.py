#!/usr/bin/env python3
"""Time the TN (weight-gradient) GEMM on the training step's shapes: dW[Mw,Nw] = dY[rows,Mw]^T X[rows,Nw]."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outfitx_amd import _lib as L

lib = L.load()
st = lambda: torch.cuda.current_stream().cuda_stream
shapes = [("Win", 3072, 1024), ("Wo", 1024, 1024), ("W1", 2048, 1024), ("W2", 1024, 2048)]
for rows in (2304, 18432):
    for name, M, N in shapes:
        A = torch.randn(rows, M, device="cuda").bfloat16(); B = torch.randn(rows, N, device="cuda").bfloat16()
        C = torch.empty(M, N, device="cuda")
        nb = lib.ofx_gemm_tn_ws(M, N, rows)
        slab = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
        def run():
            L.check(lib.ofx_gemm_tn(A.data_ptr(), M, B.data_ptr(), N, C.data_ptr(), N, M, N, rows, None, slab.data_ptr() if nb else None, nb, 1, st()))
        for _ in range(5): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print(json.dumps({"shape": name, "rows": rows, "M": M, "N": N, "splits_bytes": nb, "us": round(us, 1), "TF": round(2.0 * rows * M * N / us / 1e6, 1)}))
