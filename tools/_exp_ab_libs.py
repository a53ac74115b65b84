#!/usr/bin/env python3
"""In-process A/B of two builds of libofx_hip.so on the ViT GEMM shapes (real epilogues), alternating libraries on the
same device: settles whether an epilogue change moved the GEMM (device-to-device spread is larger than such effects)."""
import ctypes as C, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = {"old": C.CDLL(os.path.join(ROOT, "build/ab/libofx_old.so")), "new": C.CDLL(os.path.join(ROOT, "outfitx_amd/libofx_hip.so"))}
vp, i = C.c_void_p, C.c_int
for l in libs.values():
    l.ofx_gemm.restype = i
    l.ofx_gemm.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, vp]
st = lambda: torch.cuda.current_stream().cuda_stream
# (name, M, N, K, act, out_kind, resid)
shapes = [("vit qkv", 102400, 2304, 768, 0, 1, False), ("vit out", 102400, 768, 768, 0, 0, True),
          ("vit fc1", 102400, 3072, 768, 1, 1, False), ("vit fc2", 102400, 768, 3072, 0, 0, True)]
for name, M, N, K, act, ok, res in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    Cc = torch.zeros(M, N, device="cuda", dtype=torch.float32 if ok == 0 else torch.bfloat16)
    def run(lib):
        rc = lib.ofx_gemm(A.data_ptr(), W.data_ptr(), Cc.data_ptr(), bias.data_ptr(), Cc.data_ptr() if res else None, M, N, K, K, N, N if res else 0, act, ok, 1, st())
        assert rc == 0
    out = {}
    for rep in range(3):
        for tag, lib in libs.items():
            for _ in range(3): run(lib)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run(lib)
            e1.record(); torch.cuda.synchronize()
            out.setdefault(tag, []).append(round(e0.elapsed_time(e1) / 20 * 1e3, 1))
    print(json.dumps({"shape": name, **out}))
