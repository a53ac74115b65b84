#!/usr/bin/env python3
"""Is the LDS-fill rate of gemm_w2f8_kernel's slot structure a per-CU limit or a chip-wide one?  DIAG build: ablation 5 (LDS-DMA + barriers only, real tile walk),
8 (the same, every load from four L2-resident tiles) and 0 (the full kernel) with persistent grids of 256 / 128 / 64 / 32 blocks (ofx_tune(11, g)): bytes staged per
block and microsecond = the fill rate ONE CU sustains when fewer of them pull on the L2s.
    make DIAG=1 LIB=outfitx_amd/libofx_hip_diag.so OBJ=build/obj_diag && OFX_LIB=$PWD/outfitx_amd/libofx_hip_diag.so python tools/fill_grid_sweep.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outfitx_amd import _lib as L

lib = L.load(); s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda"); g.manual_seed(0)
for name, M, N, K in [("vit qkv", 102400, 2304, 768), ("vit fc2", 102400, 768, 3072)]:
    A = torch.randn(M, K, device="cuda", generator=g).half()
    Wf = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    W2 = torch.empty(N, 2 * K, device="cuda", dtype=torch.float16); L.check(lib.ofx_convert(Wf.data_ptr(), W2.data_ptr(), N, K, 3, 2, s))
    W8 = torch.empty(N, K, device="cuda", dtype=torch.uint8); sc8 = torch.empty(N, device="cuda", dtype=torch.uint8)
    L.check(lib.ofx_pack_lo8(W2.data_ptr(), W8.data_ptr(), sc8.data_ptr(), N, K, s))
    C = torch.empty(M, N, device="cuda", dtype=torch.float16)
    run = lambda: L.check(lib.ofx_gemm_w2f8(A.data_ptr(), W2.data_ptr(), W8.data_ptr(), sc8.data_ptr(), C.data_ptr(), None, None, M, N, K, K, N, 0, 0, 1, s))
    tiles = (M // 256) * (N // 256); staged = tiles * (K // 32) * 40960.0          # bytes through LDS per launch
    print(f"{name}: {tiles} tiles, {staged / 1e9:.2f} GB staged per launch", flush=True)
    for abl in (5, 8, 0):
        for grid in (256, 128, 64, 32):
            lib.ofx_tune(1, abl); lib.ofx_tune(11, grid)
            for _ in range(3): run()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
            t = float(np.median(ts))
            print(f"   ablation {abl} grid {grid:3d}: {t:8.1f} us   {staged / grid / t / 1e3:6.1f} GB/s staged per CU   {staged / t / 1e6:6.2f} TB/s chip", flush=True)
    lib.ofx_tune(1, 0); lib.ofx_tune(11, -1)
