#!/usr/bin/env python3
"""cfg1-style A/B of ofx_tune knobs on the precomputed-embedding CP forward (32 / 64 / 256 outfits, 8 of 16 items), interleaved
rounds in one process:   python tools/cfg1_ab.py KNOB V1 V2 ...   (default: knob 10 = 3 2 1 0, the split-K consumer fusions)"""
import os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); warnings.simplefilter("ignore")
from outfitx_amd import synth, _lib as L
from src.models import OutfitX
from src.models.configs import ItemEncoderConfig, OutfitXConfig
from src.models.datatypes import OutfitCompatibilityPredictionTask as CP
lib = L.load()
knob = int(sys.argv[1]) if len(sys.argv) > 1 else 10
vals = [int(v) for v in sys.argv[2:]] or [3, 2, 1, 0]
dev = torch.device("cuda", 0)
model = OutfitX(OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip")))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.outfit_transformer_weights(7).items()}, strict=False)
model = model.to(dev).eval()
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
def timeit(fn, iters=200, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters
with torch.no_grad():
    for B in (32, 64, 256):
        emb, mask = synth.outfit_batch(1235, B, 16, 8)
        e, m = cu(emb), cu(mask)
        f = lambda: model(task=CP, outfit_embedding=e, outfit_mask=m)
        r = {}
        for rnd in range(3):
            for v in vals:
                lib.ofx_tune(knob, v); r.setdefault(v, []).append(timeit(f))
        lib.ofx_tune(knob, vals[0])
        print(f"B={B}: " + "  ".join(f"knob{knob}={v}: {np.median(r[v])*1e3:.4f} ms" for v in vals), flush=True)
