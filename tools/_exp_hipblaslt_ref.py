#!/usr/bin/env python3
"""Calibration only: what the vendor library (torch.mm -> hipBLASLt) reaches on the tower GEMM shapes, plain C = A W^T
in bf16 with bf16 output (no fused epilogue).  Not used by the product path."""
import torch, json
shapes = [("vit qkv", 102400, 2304, 768), ("vit out", 102400, 768, 768), ("vit fc1", 102400, 3072, 768), ("vit fc2", 102400, 768, 3072)]
for name, M, N, K in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16()
    for _ in range(5): torch.nn.functional.linear(A, W)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): torch.nn.functional.linear(A, W)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(json.dumps({"shape": name, "us": round(us, 1), "TF": round(2.0 * M * N * K / us / 1e6, 1)}))
