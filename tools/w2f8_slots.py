#!/usr/bin/env python3
"""Where the cycles of gemm_w2f8_kernel's ping-pong slots go (DIAG build, ablation 7: s_memtime stamps around every slot, summed per wave
over all iterations of all its tiles; stamps cost an s_waitcnt lgkmcnt(0) each, so the instrumented kernel runs a few % slower).
    make DIAG=1 LIB=outfitx_amd/libofx_hip_diag.so OBJ=build/obj_diag && OFX_LIB=$PWD/outfitx_amd/libofx_hip_diag.so python tools/w2f8_slots.py
Per group (0: waves 0-3, 1: waves 4-7) and slot kind (short: s = 0..2, long: s = 3 = the iteration whose MFMA slot carries the fp8
product), average cycles per iteration of: issue + fragment reads (to lgkmcnt(0)), wait at the mid barrier, MFMA slot, wait at the end barrier."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outfitx_amd import _lib as L

lib = L.load(); s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda"); g.manual_seed(0)
grid = int(os.environ.get("OFX_GRID", "-1"))          # persistent grid size (ofx_tune 11): fewer blocks = fewer CUs in their epilogues at once
lib.ofx_tune(11, grid)
nblk = 256 if grid <= 0 else grid
# (act 1 = quick-GELU + bias: fc1's epilogue arithmetic without the LayerNorm-fold terms, which only the model path sets up)
for name, M, N, K, okind, act in [("vit fc2", 102400, 768, 3072, 1, 0), ("vit qkv", 102400, 2304, 768, 1, 0), ("vit out", 102400, 768, 768, 1, 0),
                                  ("vit fc1 plain", 102400, 3072, 768, 1, 0), ("vit fc1 quick-GELU + bias", 102400, 3072, 768, 1, 1)]:
    A = torch.randn(M, K, device="cuda", generator=g).half()
    Wf = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    W2 = torch.empty(N, 2 * K, device="cuda", dtype=torch.float16); L.check(lib.ofx_convert(Wf.data_ptr(), W2.data_ptr(), N, K, 3, 2, s))
    W8 = torch.empty(N, K, device="cuda", dtype=torch.uint8); sc8 = torch.empty(N, device="cuda", dtype=torch.uint8)
    L.check(lib.ofx_pack_lo8(W2.data_ptr(), W8.data_ptr(), sc8.data_ptr(), N, K, s))
    C = torch.empty(M, N, device="cuda", dtype=torch.float16)
    bias = torch.randn(N, device="cuda", generator=g)
    run = lambda: L.check(lib.ofx_gemm_w2f8(A.data_ptr(), W2.data_ptr(), W8.data_ptr(), sc8.data_ptr(), C.data_ptr(), bias.data_ptr() if act else None, None, M, N, K, K, N, 0, act, okind, s))
    for _ in range(30): run()
    torch.cuda.synchronize()
    dbg = torch.zeros(nblk * 32, dtype=torch.int64, device="cuda")
    lib.ofx_tune(1, 7); lib.ofx_debug_gemm_clock(dbg.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    lib.ofx_debug_gemm_clock(None); lib.ofx_tune(1, 0)
    raw = dbg.cpu().numpy().reshape(nblk, 2, 16).astype(np.float64)
    d = raw[:, :, :8].reshape(nblk, 2, 2, 4)                                # [block][group][long?][read, waitA, mfma, waitB]
    tiles = ((M + 255) // 256) * (N // 256); nk = K // 32
    it = tiles / float(nblk) * nk                                           # iterations per block (average)
    print(f"{name}: instrumented launch {e0.elapsed_time(e1) * 1e3:.0f} us; {tiles} tiles x {nk} k-steps, {it:.0f} iterations per block")
    for grp in range(2):
        for lg, nm, share in ((0, "short slots (s = 0..2)", 0.75), (1, "long slot  (s = 3)   ", 0.25)):
            v = np.median(d[:, grp, lg, :], axis=0) / (it * share)
            print(f"   group {grp} {nm}: issue+read {v[0]:6.0f}  wait@mid {v[1]:6.0f}  MFMA slot {v[2]:6.0f}  wait@end {v[3]:6.0f}   = {v.sum():6.0f} cycles per iteration")
    for grp in range(2):
        e = raw[:, grp, 8] / np.maximum(raw[:, grp, 9], 1); gp = raw[:, grp, 10] / np.maximum(raw[:, grp, 9] - 1, 1)
        print(f"   group {grp} epilogue: {np.median(e):6.0f} cycles per tile (min {e.min():.0f}, max {e.max():.0f}); epilogue end -> next tile's first barrier {np.median(gp):6.0f}")
    tot = np.median(d.sum(axis=(2, 3)), axis=0)
    print(f"   stamped cycles per block: group 0 {tot[0]:.0f}, group 1 {tot[1]:.0f}  ({tot[0] / it:.0f} per k-step)")
