#!/usr/bin/env python3
"""A few launches of one GEMM shape per kernel variant (for rocprofv3 --pmc runs)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outfitx_amd import _lib as L
lib = L.load(); s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda"); g.manual_seed(0)
M, N, K = 102400, 2304, 768
A = torch.randn(M, K, device="cuda", generator=g).bfloat16(); W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for kern in (1, 2, 3, 4):
    lib.ofx_tune(2, kern)
    for _ in range(3):
        L.check(lib.ofx_gemm(A.data_ptr(), W.data_ptr(), C.data_ptr(), None, None, M, N, K, K, N, 0, 0, 1, 1, s))
    torch.cuda.synchronize()
