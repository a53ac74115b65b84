#!/usr/bin/env python3
"""How long the HOST needs to issue one headline step (BASELINE configs[1], default schemes, towers overlapped) against how long the
device needs to run it: issue time = perf_counter around the call with an idle device, no synchronisation inside.  A step whose issue
time approaches its device time is exposed to host noise (a busy core share lengthens the step)."""
import os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); warnings.simplefilter("ignore")
from outfitx_amd import synth
from src.models import OutfitX
from src.models.configs import ItemEncoderConfig, OutfitXConfig
from src.models.datatypes import OutfitCompatibilityPredictionTask as CP
dev = torch.device("cuda", 0)
B, n = 256, 8
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
for _ in range(5): step()
torch.cuda.synchronize()
issue, total = [], []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    issue.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"10 steps back to back: the host returned after {(t1 - t0) * 1e3:.1f} ms, the device finished after {(t2 - t0) * 1e3:.1f} ms", flush=True)
print(f"host issue time per step: median {np.median(issue):.2f} ms (min {min(issue):.2f}, max {max(issue):.2f}); step incl. device: median {np.median(total):.2f} ms", flush=True)
