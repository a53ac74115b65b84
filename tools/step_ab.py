#!/usr/bin/env python3
"""A/B of an ofx_tune knob on the headline step (BASELINE configs[1]: 256 outfits x 8 items, default schemes, text tower on the side
stream), interleaved rounds in one process:   python tools/step_ab.py KNOB V1 V2 ...   e.g.  python tools/step_ab.py 11 -1 0"""
import os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); warnings.simplefilter("ignore")
from outfitx_amd import synth, _lib as L
from src.models import OutfitX
from src.models.configs import ItemEncoderConfig, OutfitXConfig
from src.models.datatypes import OutfitCompatibilityPredictionTask as CP
lib = L.load()
knob = int(sys.argv[1]); vals = [int(v) for v in sys.argv[2:]]
dev = torch.device("cuda", 0)
B, n = 256, 8
m = OutfitX(OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip")))
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.full_state_dict(7).items()}, strict=True)
m = m.to(dev).eval()
g = torch.Generator(device=dev); g.manual_seed(1236)
u8 = torch.randint(0, 256, (B, n, 3, 224, 224), generator=g, device=dev, dtype=torch.uint8)
mean = torch.tensor(synth.CLIP_MEAN, device=dev).view(1, 1, 3, 1, 1); std = torch.tensor(synth.CLIP_STD, device=dev).view(1, 1, 3, 1, 1)
px = ((u8.float() * (1 / 255.0) - mean) / std).contiguous(); del u8
ids, att = synth.token_batch(1236, B * n, 64, 8)
texts = {"input_ids": torch.from_numpy(ids).view(B, n, 64).pin_memory(), "attention_mask": torch.from_numpy(att).view(B, n, 64).pin_memory()}
mask = torch.zeros(B, n, dtype=torch.bool, device=dev)
def step():
    with torch.no_grad():
        return m(task=CP, outfit_embedding=None, outfit_mask=mask, encoder_input_dict={"images": px, "texts": texts})
for _ in range(5): step()
torch.cuda.synchronize()
res = {v: [] for v in vals}
for rnd in range(5):
    for v in vals:
        lib.ofx_tune(knob, v)
        step(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(6): step()
        torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / 6)
lib.ofx_tune(knob, vals[0])
print("  ".join(f"knob{knob}={v}: {np.median(res[v]) * 1e3:.3f} ms" for v in vals), flush=True)
for v in vals:
    print(f"   knob{knob}={v} rounds: " + " ".join(f"{t * 1e3:.2f}" for t in res[v]), flush=True)
