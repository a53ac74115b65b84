#!/usr/bin/env python3
"""Small-M GEMM plans (tile height x split-K) on the outfit transformer's shapes at 32 outfits (M = 288 rows, bf16x3: K' = 3K):
interleaved rounds, GEMM + split-K reduce timed together.  python tools/small_m_bench.py [M]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outfitx_amd import _lib as L

def main():
    lib = L.load()
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 288
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    plans = [("auto", 1)] + [(f"128x{k}", k) for k in (2, 4, 6, 8, 12, 16)] + [(f"64x{k}", 256 + k) for k in (1, 2, 4, 6, 8, 12, 16)]
    for name, N, K in (("qkv", 3072, 3072), ("out", 1024, 3072), ("fc1", 2048, 3072), ("fc2", 1024, 6144)):
        A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
        C = torch.empty(M, N, device="cuda", dtype=torch.float32)
        bias = torch.randn(N, device="cuda", generator=g)
        lib.ofx_tune(5, 2)
        slab = torch.empty(int(lib.ofx_gemm_splitk_ws(M, N, K)), dtype=torch.uint8, device="cuda")
        res = {n: [] for n, _ in plans}
        ref = None
        for rnd in range(6):
            for n, v in plans:
                lib.ofx_tune(5, v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    L.check(lib.ofx_gemm_splitk(A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), None, M, N, K, K, N, 0, 0, 0, 1, slab.data_ptr(), slab.numel(), s))
                e1.record(); e1.synchronize()
                if rnd:
                    res[n].append(e0.elapsed_time(e1) / 10)
                if rnd == 0:
                    if ref is None:
                        ref = C.clone()
                    else:
                        assert float((C - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), n
        lib.ofx_tune(5, 1)
        best = min(res, key=lambda n: np.median(res[n]))
        print(f"{name} M={M} N={N} K={K}: " + "  ".join(f"{n} {np.median(v)*1e3:5.1f}" for n, v in res.items()) + f"   best {best}", flush=True)

if __name__ == "__main__":
    main()
