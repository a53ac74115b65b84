#!/usr/bin/env python3
"""Which tile kernel is fastest for the training step's small-M GEMMs (M = 2304 rows = 256 outfits x 9)?"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outfitx_amd import _lib as L
lib = L.load()
st = lambda: torch.cuda.current_stream().cuda_stream
shapes = [("qkv", 2304, 3072, 1024), ("out/dO", 2304, 1024, 1024), ("fc1/dU", 2304, 2048, 1024), ("fc2/dH2", 2304, 1024, 2048), ("dH1", 2304, 1024, 3072),
          ("out 4352", 4352, 1024, 1024), ("fc1 4352", 4352, 2048, 1024), ("fc2 4352", 4352, 1024, 2048), ("qkv 4352", 4352, 3072, 1024), ("out 6144", 6144, 1024, 1024)]
for name, M, N, K in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    C = torch.zeros(M, N, device="cuda")
    nb = lib.ofx_gemm_splitk_ws(M, N, K)
    slab = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
    res = {}
    for tag, kind, split in (("k1", 1, 0), ("k1+splitK", 1, 1), ("k5 64x128", 5, 0), ("auto", 0, 1), ("k3", 3, 0)):
        if kind == 4 and K > 1024: continue
        lib.ofx_tune(2, kind)
        def run():
            if split and nb:
                L.check(lib.ofx_gemm_splitk(A.data_ptr(), W.data_ptr(), C.data_ptr(), None, None, M, N, K, K, N, 0, 0, 0, 1, slab.data_ptr(), nb, st()))
            else:
                L.check(lib.ofx_gemm(A.data_ptr(), W.data_ptr(), C.data_ptr(), None, None, M, N, K, K, N, 0, 0, 0, 1, st()))
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): run()
        e1.record(); torch.cuda.synchronize()
        res[tag] = round(e0.elapsed_time(e1) / 30 * 1e3, 1)
    lib.ofx_tune(2, 0)
    print(json.dumps({"shape": name, "M": M, "N": N, "K": K, "split_bytes": nb, **res}))
