#!/usr/bin/env python3
"""Three-product GEMMs on the text tower's / outfit transformer's shapes: the operand-tiles-loaded-once kernel (gemm_x3_kernel,
ofx_tune(15, 2)) against the K-concatenated single-product path (ofx_tune(15, 0)), interleaved rounds in one process, random f16 data.
    python tools/gemm_x3_bench.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outfitx_amd import _lib as L

lib = L.load(); s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda"); g.manual_seed(0)
SHAPES = [("txt qkv", 16384, 1536, 512, "f"), ("txt out", 16384, 512, 512, "r"), ("txt fc1", 16384, 2048, 512, "g"), ("txt fc2", 16384, 512, 2048, "r"),
          ("set qkv", 9216, 3072, 1024, "f"), ("set out", 9216, 1024, 1024, "r"), ("set fc1", 9216, 2048, 1024, "g"), ("set fc2", 9216, 1024, 2048, "r"),
          ("set qkv B256", 2304, 3072, 1024, "f"), ("set fc2 B256", 2304, 1024, 2048, "r")]
for name, M, N, K, ep in SHAPES:
    Af = torch.randn(M, K, device="cuda", generator=g)
    hi = Af.half(); lo = (Af - hi.float()).half()
    A3 = torch.cat([hi, lo, hi], 1).contiguous()
    Wf = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    W3 = torch.empty(N, 3 * K, device="cuda", dtype=torch.float16)
    L.check(lib.ofx_convert(Wf.data_ptr(), W3.data_ptr(), N, K, 2, 2, s))
    bias = torch.randn(N, device="cuda", generator=g)
    if ep == "r":
        C = torch.randn(M, N, device="cuda", generator=g); args = (None, C.data_ptr(), M, N, K, 3 * K, N, N, 0, 0, 2, s)
    elif ep == "g":
        C = torch.empty(M, 3 * N, device="cuda", dtype=torch.float16); args = (bias.data_ptr(), None, M, N, K, 3 * K, 3 * N, 0, 1, 2, 2, s)
    else:
        C = torch.empty(M, N, device="cuda"); args = (bias.data_ptr(), None, M, N, K, 3 * K, N, 0, 0, 0, 2, s)
    def run(knob):
        lib.ofx_tune(15, knob)
        try:
            return lib.ofx_gemm_x3(A3.data_ptr(), W3.data_ptr(), C.data_ptr(), *args)
        finally:
            lib.ofx_tune(15, 1)
    res = {2: [], 0: []}
    for rnd in range(6):
        for knob in (2, 0):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                L.check(run(knob))
            e1.record(); e1.synchronize()
            if rnd:
                res[knob].append(e0.elapsed_time(e1) / 10)
    t2, t0 = np.median(res[2]), np.median(res[0])
    print(f"{name:13s} M={M} N={N} K={K} ep={ep} | loaded-once {t2*1e3:7.1f} us {2*M*N*K/t2/1e9:6.0f} TF useful | K-concatenated {t0*1e3:7.1f} us {2*M*N*K/t0/1e9:6.0f} TF | ratio {t2/t0:.2f}", flush=True)
