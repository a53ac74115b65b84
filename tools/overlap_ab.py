#!/usr/bin/env python3
"""A/B of how the text tower shares the chip with the ViT on the headline step (BASELINE configs[1], default schemes), interleaved
rounds in one process: side stream priority (normal | low) x gemm_x3_kernel grid (persistent | one block per tile = ofx_tune(16, 0))."""
import os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); warnings.simplefilter("ignore")
from outfitx_amd import synth, _lib as L
from src.models import OutfitX
from src.models.configs import ItemEncoderConfig, OutfitXConfig
from src.models.datatypes import OutfitCompatibilityPredictionTask as CP
lib = L.load()
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
ref = step().float().cpu()
variants = [("normal", 1), ("low", 1), ("normal", 0), ("low", 0), ("off", 1)]
res = {v: [] for v in variants}
for rnd in range(5):
    for v in variants:
        m.item_encoder.overlap_towers = v[0] != "off"
        if v[0] != "off": m.item_encoder.side_stream_priority = v[0]
        lib.ofx_tune(16, v[1])
        for _ in range(2): out = step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(6): out = step()
        torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / 6)
        assert torch.equal(out.float().cpu(), ref), v
lib.ofx_tune(16, 1)
for v in variants:
    print(f"side stream {v[0]:6s} x3 grid {'persistent' if v[1] else 'per tile  '}: median {np.median(res[v]) * 1e3:.3f} ms   rounds " + " ".join(f"{t * 1e3:.2f}" for t in res[v]), flush=True)
