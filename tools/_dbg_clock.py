import os, sys, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from outfitx_amd import _lib as L
lib = L.load(); s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda"); g.manual_seed(0)
M, N, K = 102400, 768, 3072
A = torch.randn(M, K, device="cuda", generator=g).bfloat16(); W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for kern in (2, 4):
    lib.ofx_tune(2, kern)
    run = lambda: L.check(lib.ofx_gemm(A.data_ptr(), W.data_ptr(), C.data_ptr(), None, None, M, N, K, K, N, 0, 0, 1, 1, s))
    for _ in range(50): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); e1.synchronize()
    nblk = 400 * 3
    dbg = torch.zeros(nblk, 4, dtype=torch.int64, device="cuda")
    lib.ofx_debug_gemm_clock(dbg.data_ptr()); run(); torch.cuda.synchronize(); lib.ofx_debug_gemm_clock(None)
    d = dbg.cpu().numpy()
    print('kern', kern, 'kernel ms', e0.elapsed_time(e1), 'rows:', d[:3].tolist(), d[600:602].tolist(), 'median', np.median(d, 0).tolist())
