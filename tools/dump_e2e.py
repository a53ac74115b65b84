#!/usr/bin/env python3
"""Dump item embeddings + CP logits of the default scheme for one weight seed from the tree at ROOT (comparing builds):
    python tools/dump_e2e.py ROOT seed out.npz"""
import os, sys, warnings
root, ws, outp = sys.argv[1], int(sys.argv[2]), sys.argv[3]
sys.path.insert(0, root); warnings.simplefilter("ignore")
import numpy as np, torch
from outfitx_amd import synth
from src.models import OutfitX
from src.models.configs import ItemEncoderConfig, OutfitXConfig
from src.models.datatypes import OutfitCompatibilityPredictionTask as CP
B, n = 8, 8
mean = torch.tensor(synth.CLIP_MEAN).view(1, 1, 3, 1, 1); std = torch.tensor(synth.CLIP_STD).view(1, 1, 3, 1, 1)
g = torch.Generator(); g.manual_seed(9000 + ws)
u8 = torch.randint(0, 256, (B, n, 3, 224, 224), generator=g, dtype=torch.uint8)
px = ((u8.float() * (1 / 255.0) - mean) / std).contiguous()
ids, att = synth.token_batch(9000 + ws, B * n, 64, 8)
texts = {"input_ids": torch.from_numpy(ids).view(B, n, 64), "attention_mask": torch.from_numpy(att).view(B, n, 64)}
m = OutfitX(OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip")))
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.full_state_dict(ws).items()}, strict=True)
m = m.cuda().eval()
with torch.no_grad():
    emb = m.item_encoder(px.cuda(), texts)
    logit = m(task=CP, outfit_embedding=None, outfit_mask=torch.zeros(B, n, dtype=torch.bool).cuda(), encoder_input_dict={"images": px.cuda(), "texts": texts})
    logit2 = m(task=CP, outfit_embedding=emb, outfit_mask=torch.zeros(B, n, dtype=torch.bool).cuda())
np.savez(outp, emb=emb.cpu().numpy(), logit=logit.cpu().numpy(), logit_from_emb=logit2.cpu().numpy())
print("dumped", outp, float(logit.abs().max()))
