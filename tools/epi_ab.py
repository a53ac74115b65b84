#!/usr/bin/env python3
"""A/B of gemm_w2f8_kernel's operand-type epilogue forms on the ViT's f16-output shapes (ofx_tune(18, v): 0 through LDS, 1 straight from the accumulator
layout as 16 rows x 64 B buffer stores, 2 the same as 8 rows x 128 B global stores), interleaved rounds in one process, random f16 data.
    python tools/epi_ab.py [modes, default 1,2]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outfitx_amd import _lib as L

lib = L.load(); s = torch.cuda.current_stream().cuda_stream
modes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,2").split(",")]
skews = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]          # ofx_tune(19, v): start skew by XCD (KArgs::skew in gemm_w2f8.hip)
modes = [(m, k) for k in skews for m in modes]
g = torch.Generator(device="cuda"); g.manual_seed(0)
for name, M, N, K, act in [("vit qkv", 102400, 2304, 768, 0), ("vit fc1 plain", 102400, 3072, 768, 0), ("vit fc1 quick-GELU + bias", 102400, 3072, 768, 1)]:
    A = torch.randn(M, K, device="cuda", generator=g).half()
    Wf = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    W2 = torch.empty(N, 2 * K, device="cuda", dtype=torch.float16); L.check(lib.ofx_convert(Wf.data_ptr(), W2.data_ptr(), N, K, 3, 2, s))
    W8 = torch.empty(N, K, device="cuda", dtype=torch.uint8); sc8 = torch.empty(N, device="cuda", dtype=torch.uint8)
    L.check(lib.ofx_pack_lo8(W2.data_ptr(), W8.data_ptr(), sc8.data_ptr(), N, K, s))
    C = torch.empty(M, N, device="cuda", dtype=torch.float16)
    bias = torch.randn(N, device="cuda", generator=g)
    run = lambda: L.check(lib.ofx_gemm_w2f8(A.data_ptr(), W2.data_ptr(), W8.data_ptr(), sc8.data_ptr(), C.data_ptr(), bias.data_ptr() if act else None, None, M, N, K, K, N, 0, act, 1, s))
    for _ in range(20): run()
    t = {m: [] for m in modes}; ref = None
    for rnd in range(6):
        for m in modes:
            lib.ofx_tune(18, m[0]); lib.ofx_tune(19, m[1])
            run(); torch.cuda.synchronize()
            if rnd == 0:
                if ref is None: ref = C.clone()
                else: assert torch.equal(ref, C), f"mode {m} differs"
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            t[m].append(e0.elapsed_time(e1) / 20 * 1e3)
    lib.ofx_tune(18, 1); lib.ofx_tune(19, 0)
    print(f"{name}:\n" + "\n".join(f"   epilogue {m[0]} skew {m[1]:3d}: median {np.median(t[m]):6.1f} us (min {min(t[m]):6.1f})" for m in modes), flush=True)
