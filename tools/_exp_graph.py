#!/usr/bin/env python3
"""Does the small-batch CP forward (BASELINE configs[0]: 32 outfits x 16 padded slots, precomputed embeddings) replay from a
HIP graph captured through torch.cuda.graph, and what does it save?  Launches go to torch's current stream, so stream capture
sees them."""
import os, sys, time, json, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); warnings.simplefilter("ignore")
import torch, numpy as np
from outfitx_amd import synth
from src.models import OutfitX
from src.models.configs import ItemEncoderConfig, OutfitXConfig
from src.models.datatypes import OutfitCompatibilityPredictionTask as CP
dev = torch.device("cuda")
m = OutfitX(OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip")))
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.outfit_transformer_weights(7).items()}, strict=False)
m = m.to(dev).eval()
res = {}
for B in (32, 256):
    emb_np, mask_np = synth.outfit_batch(1234, B, 16, 8) if hasattr(synth, "outfit_batch") else (np.random.default_rng(0).standard_normal((B, 16, 1024), dtype=np.float32), np.arange(16)[None, :].repeat(B, 0) >= 8)
    x = torch.from_numpy(np.ascontiguousarray(emb_np)).to(dev); mask = torch.from_numpy(np.ascontiguousarray(mask_np)).to(dev)
    def fwd():
        with torch.no_grad():
            return m(task=CP, outfit_embedding=x, outfit_mask=mask)
    for _ in range(3): ref = fwd()
    torch.cuda.synchronize()
    def timed(fn, n=50):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    res[f"B{B}_eager_ms"] = timed(fwd)
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2): fwd()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            out = fwd()
        g.replay(); torch.cuda.synchronize()
        res[f"B{B}_graph_equal"] = bool(torch.equal(out, ref))
        res[f"B{B}_graph_ms"] = timed(g.replay)
        x.mul_(0.5); g.replay(); torch.cuda.synchronize()          # new contents in the same buffers
        res[f"B{B}_graph_tracks_inputs"] = bool(torch.equal(out, fwd()))
        x.mul_(2.0)
    except Exception as e:
        res[f"B{B}_graph_error"] = repr(e)[:300]
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items()}))
