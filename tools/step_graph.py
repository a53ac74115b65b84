#!/usr/bin/env python3
"""The headline step (BASELINE configs[1], default schemes, text tower on the side stream) captured into ONE HIP graph and replayed against
the same step issued launch by launch: per-step device times (stream markers), equality of the logits."""
import os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); warnings.simplefilter("ignore")
from outfitx_amd import synth
from src.models import OutfitX
from src.models.configs import ItemEncoderConfig, OutfitXConfig
from src.models.datatypes import OutfitCompatibilityPredictionTask as CP
dev = torch.device("cuda", 0)
B, n, K = 256, 8, 30
m = OutfitX(OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip")))
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.full_state_dict(7).items()}, strict=True)
m = m.to(dev).eval()
px, ids, att = synth.bench_batch(1236, B, n)
px = torch.from_numpy(px).to(dev)
texts = {"input_ids": torch.from_numpy(ids).view(B, n, 64).pin_memory(), "attention_mask": torch.from_numpy(att).view(B, n, 64).pin_memory()}
mask = torch.zeros(B, n, dtype=torch.bool, device=dev)
def step():
    with torch.no_grad():
        return m(task=CP, outfit_embedding=None, outfit_mask=mask, encoder_input_dict={"images": px, "texts": texts})
for _ in range(5): ref = step()
torch.cuda.synchronize()
def timed(fn, K):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    torch.cuda.synchronize(); t0 = time.perf_counter(); ev[0].record()
    for i in range(K):
        out = fn(); ev[i + 1].record()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / K * 1e3
    per = [ev[i].elapsed_time(ev[i + 1]) for i in range(K)]
    return out, wall, per
out, wall, per = timed(step, K)
print(f"launch by launch: {wall:.3f} ms per step; median {np.median(per):.3f}, steps over median + 0.5 ms: {sum(p > np.median(per) + 0.5 for p in per)} of {K}; " + " ".join(f"{p:.1f}" for p in per), flush=True)
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream(dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    for _ in range(2): step()                       # warm the capture stream's allocator pool
torch.cuda.current_stream(dev).wait_stream(side); torch.cuda.synchronize()
with torch.cuda.graph(g, stream=side):
    gout = step()
torch.cuda.synchronize()
def replay():
    g.replay(); return gout
out2, wall2, per2 = timed(replay, K)
print(f"one graph launch: {wall2:.3f} ms per step; median {np.median(per2):.3f}, steps over median + 0.5 ms: {sum(p > np.median(per2) + 0.5 for p in per2)} of {K}; " + " ".join(f"{p:.1f}" for p in per2), flush=True)
print("logits equal:", bool(torch.equal(out2, ref)), " max |d| =", float((out2 - ref).abs().max()), flush=True)
