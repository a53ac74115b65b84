#!/usr/bin/env python3
"""Would a head-major q|k|v layout (each (sequence, head) tile contiguous) make the ViT attention faster than the row-major
[rows, 2304] layout the QKV GEMM writes?  Emulated with the op-level call: n_head = 1, ld = 64, k / v as separate arrays."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outfitx_amd import _lib as L
lib = L.load()
st = lambda: torch.cuda.current_stream().cuda_stream
N, S, H = 2048, 50, 12
def t(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
qkv = torch.randn(N * S, 3 * H * 64, device="cuda").bfloat16(); out = torch.empty(N * S, H * 64, device="cuda", dtype=torch.bfloat16)
row_major = t(lambda: L.check(lib.ofx_attention(qkv.data_ptr(), out.data_ptr(), None, N, S, H, 3 * H * 64, H * 64, H * 64, 2 * H * 64, 0, 0, 0.125, 1, st())))
n2 = N * H
flat = torch.randn(3 * n2 * S * 64, device="cuda").bfloat16(); out2 = torch.empty(n2 * S, 64, device="cuda", dtype=torch.bfloat16)
head_major = t(lambda: L.check(lib.ofx_attention(flat.data_ptr(), out2.data_ptr(), None, n2, S, 1, 64, 64, n2 * S * 64, 2 * n2 * S * 64, 0, 0, 0.125, 1, st())))
print(json.dumps({"row_major_us": round(row_major, 1), "head_major_us": round(head_major, 1)}))
