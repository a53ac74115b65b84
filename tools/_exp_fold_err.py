#!/usr/bin/env python3
"""Tower error vs the reference goldens with the LayerNorm folding on / off (diagnostic)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outfitx_amd import synth, _lib as L
from src.models import OutfitX
from src.models.configs import ItemEncoderConfig, OutfitXConfig
lib = L.load()
m = OutfitX(OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip")))
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.full_state_dict(7).items()}, strict=True)
m = m.cuda().eval()
G = lambda n: np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden", n + ".npz"))
rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
gv, gt = G("vit_n4"), G("text_n8")
px = torch.from_numpy(synth.pixel_values(1237, 4)).view(4, 1, 3, 224, 224).cuda()
ids, att = synth.token_batch(1238, 8, 64, gt["n_real"])
tok = {"input_ids": torch.from_numpy(ids).view(8, 1, 64), "attention_mask": torch.from_numpy(att).view(8, 1, 64)}
for prec in ("bf16", "f16"):
    m.item_encoder.set_precision(prec)
    for fold in (0, 1):
        lib.ofx_tune(6, fold)
        with torch.no_grad():
            v = m.item_encoder.image_enc(px, normalize=False).view(4, 512).cpu().numpy()
            t = m.item_encoder.text_enc(tok, normalize=False).view(8, 512).cpu().numpy()
        print(prec, "fold", fold, "vit err %.2e" % rel(v, gv["image_embeds"]), "text err %.2e" % rel(t, gt["text_embeds"]))
