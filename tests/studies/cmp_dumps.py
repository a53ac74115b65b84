#!/usr/bin/env python3
"""Compare two dump_e2e.py outputs (and both against the fp32 oracle): python tests/studies/cmp_dumps.py a.npz b.npz seed"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from outfitx_amd import synth
from oracle import np_oracle as O
a, b, ws = np.load(sys.argv[1]), np.load(sys.argv[2]), int(sys.argv[3])
rel = lambda x, y: float(np.abs(x - y).max() / np.abs(y).max())
B, n = 8, 8
mean = torch.tensor(synth.CLIP_MEAN).view(1, 1, 3, 1, 1); std = torch.tensor(synth.CLIP_STD).view(1, 1, 3, 1, 1)
g = torch.Generator(); g.manual_seed(9000 + ws)
u8 = torch.randint(0, 256, (B, n, 3, 224, 224), generator=g, dtype=torch.uint8)
px = ((u8.float() * (1 / 255.0) - mean) / std).contiguous().numpy()
ids, att = synth.token_batch(9000 + ws, B * n, 64, 8)
emb = O.item_encoder(px, ids.reshape(B, n, 64), att.reshape(B, n, 64), synth.vision_weights(ws), synth.text_weights(ws))
ref = O.cp_forward(emb, np.zeros((B, n), bool), synth.outfit_transformer_weights(ws))
for name, d in (("A", a), ("B", b)):
    print(name, "img emb err", rel(d["emb"][..., :512], emb[..., :512]), "txt emb err", rel(d["emb"][..., 512:], emb[..., 512:]),
          "logit err", rel(d["logit"], ref), "logit(from its emb, bf16x3 set) err", rel(d["logit_from_emb"], ref))
print("A vs B: img emb", rel(a["emb"][..., :512], b["emb"][..., :512]), "txt emb", rel(a["emb"][..., 512:], b["emb"][..., 512:]), "logit", rel(a["logit"], b["logit"]))
