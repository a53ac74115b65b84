#!/usr/bin/env python3
"""Text tower on 2048 item texts drawn from 132 distinct category strings (Polyvore's categories.json): plain vs dedup_texts."""
import os, sys, time, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outfitx_amd import synth
from src.models import OutfitX
from src.models.configs import ItemEncoderConfig, OutfitXConfig
m = OutfitX(OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip"))).cuda().eval()
enc = m.item_encoder.text_enc
ids, att = synth.token_batch(5, 132, 64, synth.ragged_lengths(5, 132, 3, 8))
pick = np.random.default_rng(0).integers(0, 132, 2048)
tok = {"input_ids": torch.from_numpy(ids[pick]).view(2048, 1, 64).pin_memory(), "attention_mask": torch.from_numpy(att[pick]).view(2048, 1, 64).pin_memory()}
def t(n=10):
    with torch.no_grad():
        for _ in range(2): enc(tok)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): enc(tok)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
res = {"texts": 2048, "distinct": 132, "ms_plain": round(t(), 3)}
enc.dedup_texts = True
res["ms_dedup"] = round(t(), 3)
print(json.dumps(res))
