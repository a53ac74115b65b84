#!/usr/bin/env python3
"""cfg1 (32 precomputed outfits) forward repeated: target for `rocprofv3 --kernel-trace --stats`."""
import os, sys, warnings
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
    for _ in range(200):
        m(task=CP, outfit_embedding=e, outfit_mask=k)
torch.cuda.synchronize()
