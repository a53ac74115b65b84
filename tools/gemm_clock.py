#!/usr/bin/env python3
"""In-kernel clock and cycles per k-tile of the big-tile GEMM main loop (diagnostic build path: the stamps only
execute when ofx_debug_gemm_clock() armed a buffer).  Random data, after a 2 s warm-up of back-to-back launches."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outfitx_amd import _lib as L
lib = L.load(); s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda"); g.manual_seed(0)
for name, M, N, K, kern, abl in [("vit fc2", 102400, 768, 3072, 2, 0), ("vit fc2", 102400, 768, 3072, 4, 0), ("vit qkv", 102400, 2304, 768, 2, 0), ("vit qkv", 102400, 2304, 768, 4, 0)]:
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16(); W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    lib.ofx_tune(2, kern); lib.ofx_tune(1, abl)
    run = lambda: L.check(lib.ofx_gemm(A.data_ptr(), W.data_ptr(), C.data_ptr(), None, None, M, N, K, K, N, 0, 0, 1, 1, s))
    t0 = time.time()
    while time.time() - t0 < 2.0:
        for _ in range(20): run()
        torch.cuda.synchronize()
    nblk = ((M + 255) // 256) * (N // (256 if kern == 2 else 128))
    dbg = torch.zeros(nblk, 4, dtype=torch.int64, device="cuda")
    lib.ofx_debug_gemm_clock(dbg.data_ptr()); run(); torch.cuda.synchronize(); lib.ofx_debug_gemm_clock(None)
    d = dbg.cpu().numpy().astype(np.float64)
    clk = np.median(d[:, 0] / d[:, 1]) * 0.1   # GHz
    cyc = np.median(d[:, 0]); nk = K // 64
    mfma_cyc = 64 * 16 * 2                      # two waves per SIMD, 64 MFMAs of 16 cycles each, per k-tile
    pro, epi = np.median(d[:, 2]), np.median(d[:, 3])
    print(f"{name} kern {kern} abl {abl}: prologue {pro:.0f} cyc, loop {cyc:.0f}, epilogue(incl. store drain) {epi:.0f}; clock {clk:.2f} GHz, main loop {cyc:.0f} cycles = {cyc/nk:.0f} per k-tile (MFMA-bound floor {mfma_cyc}), peak at this clock {clk/2.4*2500:.0f} TF", flush=True)
lib.ofx_tune(2, 0); lib.ofx_tune(1, 0)
