#!/usr/bin/env python3
"""End-to-end CP-logit error of the default tower scheme vs the fp32 oracle under the library's knobs (LayerNorm folding, fused
QKV + attention, tower scheme), per weight seed: which implementation choice moves the error.  python tests/studies/e2e_knobs_gpu.py 4 6"""
import json, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); warnings.simplefilter("ignore")
import numpy as np, torch
from outfitx_amd import synth, _lib as L
from oracle import np_oracle as O
from src.models import OutfitX
from src.models.configs import ItemEncoderConfig, OutfitXConfig
from src.models.datatypes import OutfitCompatibilityPredictionTask as CP

lib = L.load()
seeds = [int(a) for a in sys.argv[1:]] or [4, 6]
B, n = 8, 8
mean = torch.tensor(synth.CLIP_MEAN).view(1, 1, 3, 1, 1); std = torch.tensor(synth.CLIP_STD).view(1, 1, 3, 1, 1)
variants = [("default", "f16w2", {}), ("fold=1", "f16w2", {6: 1}), ("fold=0", "f16w2", {6: 0}), ("fuse=0", "f16w2", {9: 0}), ("fold=0,fuse=0", "f16w2", {6: 0, 9: 0}), ("prune_q=0", "f16w2", {8: 0}), ("splitk=0", "f16w2", {5: 0}),
            ("f16 single", "f16", {}), ("f16x3 (all tower GEMMs three-product)", "f16x3", {})]
for ws in seeds:
    g = torch.Generator(); g.manual_seed(9000 + ws)
    u8 = torch.randint(0, 256, (B, n, 3, 224, 224), generator=g, dtype=torch.uint8)
    px = ((u8.float() * (1 / 255.0) - mean) / std).contiguous()
    ids, att = synth.token_batch(9000 + ws, B * n, 64, 8)
    texts = {"input_ids": torch.from_numpy(ids).view(B, n, 64), "attention_mask": torch.from_numpy(att).view(B, n, 64)}
    mask = np.zeros((B, n), bool)
    ref = O.cp_forward(O.item_encoder(px.numpy(), ids.reshape(B, n, 64), att.reshape(B, n, 64), synth.vision_weights(ws), synth.text_weights(ws)), mask,
                       synth.outfit_transformer_weights(ws))
    row = {}
    sd = {k: torch.from_numpy(v) for k, v in synth.full_state_dict(ws).items()}
    for name, tp, knobs in variants:
        if tp.partition('@')[0] not in L.TOWER_SCHEMES:
            continue
        m = OutfitX(OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip")), tower_precision=tp)
        m.load_state_dict(sd, strict=True); m = m.cuda().eval()
        for k, v in knobs.items():
            lib.ofx_tune(k, v)
        try:
            with torch.no_grad():
                got = m(task=CP, outfit_embedding=None, outfit_mask=torch.from_numpy(mask).cuda(), encoder_input_dict={"images": px.cuda(), "texts": texts}).cpu().numpy()
        finally:
            lib.ofx_tune(6, 2); lib.ofx_tune(9, 1); lib.ofx_tune(8, 1); lib.ofx_tune(5, 1)
        row[name] = float(np.abs(got - ref).max() / np.abs(ref).max())
        del m; torch.cuda.empty_cache()
    print(json.dumps({"weight_seed": ws, **{k: float(f"{v:.3g}") for k, v in row.items()}}), flush=True)
