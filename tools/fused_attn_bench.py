#!/usr/bin/env python3
"""ViT attention block: fused QKV-projection + attention kernel vs the unfused pair (QKV GEMM -> HBM -> attention kernel), interleaved
rounds in one process, random f16 data, with and without the LayerNorm-fold epilogue.   python tools/fused_attn_bench.py [n_img]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outfitx_amd import _lib as L

def main():
    lib = L.load()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    S, H, W = 50, 12, 768
    rows = n * S
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    X = torch.randn(rows, W, device="cuda", generator=g).half()
    Wq = (torch.randn(3 * W, W, device="cuda", generator=g) / W ** 0.5).half()
    bias = torch.randn(3 * W, device="cuda", generator=g) * 0.1
    stat = torch.stack([torch.randn(rows, device="cuda", generator=g) * 0.05, 1 + 0.1 * torch.randn(rows, device="cuda", generator=g)], 1).contiguous()
    cs = torch.randn(3 * W, device="cuda", generator=g) * 0.2
    qkv = torch.empty(rows, 3 * W, device="cuda", dtype=torch.float16)
    out = torch.empty(rows, W, device="cuda", dtype=torch.float16)
    def fused(): L.check(lib.ofx_fused_qkv_attention(X.data_ptr(), Wq.data_ptr(), bias.data_ptr(), stat.data_ptr(), cs.data_ptr(), out.data_ptr(), n, S, W, H, W, W, 0.125, 2, s))
    def pair():
        L.check(lib.ofx_gemm(X.data_ptr(), Wq.data_ptr(), qkv.data_ptr(), bias.data_ptr(), None, rows, 3 * W, W, W, 3 * W, 0, 0, 1, 2, s))
        L.check(lib.ofx_attention(qkv.data_ptr(), out.data_ptr(), None, n, S, H, 3 * W, W, W, 2 * W, 0, 0, 0.125, 2, s))
    res = {"fused": [], "pair": []}
    for rnd in range(7):
        for k, fn in (("fused", fused), ("pair", pair)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); e1.synchronize()
            if rnd:
                res[k].append(e0.elapsed_time(e1) / 5)
    fl = 2.0 * rows * 3 * W * W
    for k, v in res.items():
        t = np.median(v)
        print(f"{k:6s} {t * 1e3:8.1f} us per ViT layer ({n} images)   QKV GEMM flops / time = {fl / t / 1e9:6.0f} TF", flush=True)

if __name__ == "__main__":
    main()
