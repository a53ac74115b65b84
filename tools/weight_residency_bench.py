#!/usr/bin/env python3
"""Does the small-M GEMM's weight stream care whether the cycled weight set fits the 256 MiB Infinity Cache?  One shape, NB distinct
weight buffers visited round-robin (total = NB x 18.9 MB), GEMM + split-K reduce timed together."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outfitx_amd import _lib as L
lib = L.load()
s = torch.cuda.current_stream().cuda_stream
M, N, K = 288, 3072, 3072
A = torch.randn(M, K, device="cuda").bfloat16()
C = torch.empty(M, N, device="cuda", dtype=torch.float32)
bias = torch.randn(N, device="cuda")
slab = torch.empty(int(lib.ofx_gemm_splitk_ws(M, N, K)), dtype=torch.uint8, device="cuda")
for NB in (1, 4, 8, 12, 16, 24, 32):
    Ws = [(torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16() for _ in range(NB)]
    def run(reps):
        for r in range(reps):
            for W in Ws:
                L.check(lib.ofx_gemm_splitk(A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), None, M, N, K, K, N, 0, 0, 0, 1, slab.data_ptr(), slab.numel(), s))
    reps = max(2, 96 // NB)
    run(reps); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(reps); e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) / (reps * NB)
    print(f"NB={NB:3d} total {NB * N * K * 2 / 1e6:7.1f} MB: {t * 1e3:6.1f} us per GEMM+reduce = {N * K * 2 / t / 1e9:5.2f} TB/s of weights", flush=True)
    del Ws
