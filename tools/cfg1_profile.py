#!/usr/bin/env python3
"""BASELINE configs[0] on the GPU (32 precomputed-embedding outfits, 8 of 16 items): 50 CP forwards for `rocprofv3 --kernel-trace --stats`.
    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 /root/repo/tools/cfg1_profile.py"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); warnings.simplefilter("ignore")
import numpy as np, torch
from outfitx_amd import synth
from src.models import OutfitX
from src.models.configs import ItemEncoderConfig, OutfitXConfig
from src.models.datatypes import OutfitCompatibilityPredictionTask as CP
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = OutfitX(OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip")))
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.outfit_transformer_weights(7).items()}, strict=False)
m = m.cuda().eval()
emb, mask = synth.outfit_batch(1235, B, 16, 8)
e, k = torch.from_numpy(emb).cuda(), torch.from_numpy(mask).cuda()
with torch.no_grad():
    for _ in range(5):
        m(task=CP, outfit_embedding=e, outfit_mask=k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        m(task=CP, outfit_embedding=e, outfit_mask=k)
    torch.cuda.synchronize()
print(f"cfg1 B={B}: {(time.perf_counter() - t0) / 50 * 1e3:.4f} ms per forward")
