import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outfitx_amd import synth, _lib as L
from src.models import OutfitX
from src.models.configs import ItemEncoderConfig, OutfitXConfig
from src.models.datatypes import OutfitCompatibilityPredictionTask as CP
lib = L.load()
cfg = OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip")); cfg.transformer.dropout = 0.0
m = OutfitX(cfg, train_precision="bf16", precision="bf16")
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.full_state_dict(7).items()}, strict=True)
m = m.cuda()
B = 256
emb, mask = synth.outfit_batch(2024, B, 16, synth.ragged_lengths(2024, B, 1, 16))
x, k = torch.from_numpy(emb).cuda(), torch.from_numpy(mask).cuda()
def fwd(sl, train):
    m.train(train)
    if train:
        return m(task=CP, outfit_embedding=x[sl], outfit_mask=k[sl]).detach().squeeze(-1)
    with torch.no_grad():
        return m(task=CP, outfit_embedding=x[sl], outfit_mask=k[sl]).squeeze(-1)
for train in (False, True):
    for kind in (0, 1, 2, 3):
        for sk in (1, 0):
            lib.ofx_tune(2, kind); lib.ofx_tune(5, sk)
            yf = fwd(slice(0, B), train); yh = torch.cat([fwd(slice(0, B // 2), train), fwd(slice(B // 2, B), train)])
            print("train" if train else "score", "kind", kind, "splitk", sk, "max |full - halves| = %.3e" % float((yf - yh).abs().max()))
lib.ofx_tune(2, 0); lib.ofx_tune(5, 1)
