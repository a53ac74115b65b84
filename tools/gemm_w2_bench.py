#!/usr/bin/env python3
"""Split-weight GEMM micro-benchmark on the ViT shapes: single product vs the dual-weight kernel vs the 128x128 kernel with a
wrapping A index, interleaved rounds in one process, random f16 data.   python tools/gemm_w2_bench.py [name-filter ...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outfitx_amd import _lib as L

SHAPES = [("vit out", 102400, 768, 768, "r"), ("vit fc2", 102400, 768, 3072, "r"), ("patch", 100352, 768, 3072, "f"),
          ("vit qkv", 102400, 2304, 768, "b"), ("vit fc1", 102400, 3072, 768, "g"),
          # the text tower's three-product GEMMs as dual-weight calls on K' = 2 K (A' = [hi | lo], W rows [hi | hi] / [lo | 0]): name them to run them
          ("txt qkv", 16384, 1536, 1024, "f"), ("txt out", 16384, 512, 1024, "r"), ("txt fc1", 16384, 2048, 1024, "g"), ("txt fc2", 16384, 512, 4096, "r")]


def main():
    lib = L.load()
    only = sys.argv[1:]
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for name, M, N, K, ep in SHAPES:
        if (only and not any(o in name for o in only)) or (not only and name.startswith("txt")):
            continue
        lib.ofx_tune(2, 6 if name.startswith("txt") else 0)          # small grids: force the dual-weight 256x256 kernel
        A = torch.randn(M, K, device="cuda", generator=g).half()
        Wf = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
        data = os.environ.get("OFX_DATA", "randn")          # operand values: the kernels' clocks follow what the matrix pipe toggles
        if data == "zero": A.zero_(); Wf.zero_()
        elif data == "azero": A.zero_()
        elif data == "const": A.fill_(1.0); Wf.fill_(0.03125)
        W = Wf.half()
        W2 = torch.empty(N, 2 * K, device="cuda", dtype=torch.float16)
        L.check(lib.ofx_convert(Wf.data_ptr(), W2.data_ptr(), N, K, 3, 2, s))
        W8 = torch.empty(N, K, device="cuda", dtype=torch.uint8); sc8 = torch.empty(N, device="cuda", dtype=torch.uint8)
        f8 = K % 128 == 0 and N % 128 == 0
        if f8:
            L.check(lib.ofx_pack_lo8(W2.data_ptr(), W8.data_ptr(), sc8.data_ptr(), N, K, s))
        C = torch.empty(M, N, device="cuda", dtype=torch.float16 if ep in "bg" else torch.float32)
        if ep == "r":
            C.normal_(generator=g)
        bias = torch.randn(N, device="cuda", generator=g)
        resid = C.data_ptr() if ep == "r" else None
        act, okind = (1 if ep == "g" else 0), (1 if ep in "bg" else 0)
        abl = [int(v) for v in os.environ.get("OFX_W2_ABLATE", "").split(",") if v]
        def ablated(a):
            def f():
                lib.ofx_tune(1, a)
                try:
                    return lib.ofx_gemm_w2(A.data_ptr(), W2.data_ptr(), C.data_ptr(), bias.data_ptr(), resid, M, N, K, K, N, N, act, okind, 2, s)
                finally:
                    lib.ofx_tune(1, 0)
            return f
        runs = {"x1": lambda: lib.ofx_gemm(A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), resid, M, N, K, K, N, N, act, okind, 2, s),
                "w2": lambda: lib.ofx_gemm_w2(A.data_ptr(), W2.data_ptr(), C.data_ptr(), bias.data_ptr(), resid, M, N, K, K, N, N, act, okind, 2, s)}
        if f8:         # the fp8 correction product (gemm_w2f8_kernel); OFX_F8_ABLATE=1,2,3,4 adds its DIAG ablations
            runs["w2f8"] = lambda: lib.ofx_gemm_w2f8(A.data_ptr(), W2.data_ptr(), W8.data_ptr(), sc8.data_ptr(), C.data_ptr(), bias.data_ptr(), resid, M, N, K, K, N, N, act, okind, s)
            def f8_ablated(a):
                def f():
                    lib.ofx_tune(1, a)
                    try:
                        return runs["w2f8"]()
                    finally:
                        lib.ofx_tune(1, 0)
                return f
            for a_ in [int(v) for v in os.environ.get("OFX_F8_ABLATE", "").split(",") if v]:
                runs[f"w2f8 abl{a_}"] = f8_ablated(a_)
        for a_ in abl:
            runs[f"abl{a_}"] = ablated(a_)
        if os.environ.get("OFX_KNOB"):                  # A/B of an ofx_tune knob on the dual-weight kernel: OFX_KNOB=11:0:256 (knob:value:restore)
            kn, kv, kr = [int(v) for v in os.environ["OFX_KNOB"].split(":")]
            def knobbed():
                lib.ofx_tune(kn, kv)
                try:
                    return runs["w2"]()
                finally:
                    lib.ofx_tune(kn, kr)
            runs[f"w2 knob{kn}={kv}"] = knobbed
        def x1_ablated(a):
            def f():
                lib.ofx_tune(1, a)
                try:
                    return runs["x1"]()
                finally:
                    lib.ofx_tune(1, 0)
            return f
        for a_ in [int(v) for v in os.environ.get("OFX_X1_ABLATE", "").split(",") if v]:        # single-product kernel's ablations (DIAG build)
            runs[f"x1 abl{a_}"] = x1_ablated(a_)
        res = {k: [] for k in runs}
        for rnd in range(6):
            for k, fn in runs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    L.check(fn())
                e1.record(); e1.synchronize()
                if rnd:
                    res[k].append(e0.elapsed_time(e1) / 5)
        t1, t2 = np.median(res["x1"]), np.median(res["w2"])
        print(f"{name:8s} M={M} N={N} K={K} ep={ep} | x1 {t1*1e3:7.1f} us {2*M*N*K/t1/1e9:6.0f} TF | w2 {t2*1e3:7.1f} us "
              f"{2*M*N*K/t2/1e9:6.0f} TF useful, {4*M*N*K/t2/1e9:6.0f} TF executed | w2/x1 {t2/t1:.2f}"
              + "".join(f" | {k} {np.median(v)*1e3:7.1f} us" for k, v in res.items() if k not in ("x1", "w2")), flush=True)


if __name__ == "__main__":
    main()
