#!/usr/bin/env python3
"""End-to-end CP-logit error of the bench workload against the fp32 numpy oracle, over several independent batches, with the
towers' LayerNorms folded (default) and materialised, bf16 and f16 towers: is the bench line's single 8-outfit number a draw of
rounding noise or a systematic cost of folding?   python tests/studies/e2e_parity.py [batches] [outfits_per_batch]"""
import json, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); warnings.simplefilter("ignore")
import numpy as np, torch
from outfitx_amd import synth, _lib as L
from oracle import np_oracle as O
from src.models import OutfitX
from src.models.configs import ItemEncoderConfig, OutfitXConfig
from src.models.datatypes import OutfitCompatibilityPredictionTask as CP

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
WS = int(sys.argv[3]) if len(sys.argv) > 3 else 7          # weight seed: the weight-rounding part of the error is one draw per seed
n = 8
dev = torch.device("cuda")
lib = L.load()
Wt, Wv, Wx = synth.outfit_transformer_weights(WS), synth.vision_weights(WS), synth.text_weights(WS)
models = {}
for tp in ("bf16", "f16"):
    m = OutfitX(OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip")), precision="bf16x3", tower_precision=tp)
    m.load_state_dict({kk: torch.from_numpy(v) for kk, v in synth.full_state_dict(WS).items()}, strict=True)
    models[tp] = m.to(dev).eval()
mean = torch.tensor(synth.CLIP_MEAN).view(1, 1, 3, 1, 1); std = torch.tensor(synth.CLIP_STD).view(1, 1, 3, 1, 1)
rows = []
for b in range(nb):
    g = torch.Generator(); g.manual_seed(9000 + b)
    u8 = torch.randint(0, 256, (k, n, 3, 224, 224), generator=g, dtype=torch.uint8)
    px = ((u8.float() * (1 / 255.0) - mean) / std).contiguous()
    ids, att = synth.token_batch(9000 + b, k * n, 64, 8)
    mask = np.zeros((k, n), bool)
    t0 = time.perf_counter()
    emb = O.item_encoder(px.numpy(), ids.reshape(k, n, 64), att.reshape(k, n, 64), Wv, Wx)
    ref = O.cp_forward(emb, mask, Wt)
    texts = {"input_ids": torch.from_numpy(ids).view(k, n, 64), "attention_mask": torch.from_numpy(att).view(k, n, 64)}
    for tp in ("bf16", "f16"):
        for fold in (2, 1, 0):
            lib.ofx_tune(6, fold)
            with torch.no_grad():
                got = models[tp](task=CP, outfit_embedding=None, outfit_mask=torch.from_numpy(mask).to(dev),
                                 encoder_input_dict={"images": px.to(dev), "texts": texts}).float().cpu().numpy()
            lib.ofx_tune(6, 2)
            d = got.reshape(-1) - ref.reshape(-1)
            rows.append({"batch": b, "towers": tp, "fold": fold, "max_rel": float(np.abs(d).max() / np.abs(ref).max()),
                         "rms_rel": float(np.sqrt((d ** 2).mean()) / np.sqrt((ref ** 2).mean()))})
    print(f"batch {b}: oracle {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
summary = {}
for tp in ("bf16", "f16"):
    for fold in (2, 1, 0):
        r = [x for x in rows if x["towers"] == tp and x["fold"] == fold]
        summary[f"{tp}_fold{fold}"] = {"max_rel_per_batch": [round(x["max_rel"], 5) for x in r],
                                      "mean_of_max_rel": round(float(np.mean([x["max_rel"] for x in r])), 5),
                                      "mean_rms_rel": round(float(np.mean([x["rms_rel"] for x in r])), 5)}
print(json.dumps({"weight_seed": WS, "batches": nb, "outfits_per_batch": k, "metric": "max|d| / max|ref| and rms(d) / rms(ref) of CP logits vs the fp32 oracle", **summary}))
