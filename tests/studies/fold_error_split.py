#!/usr/bin/env python3
"""Tower-level error of the LayerNorm-folded vs materialised path against the fp32 oracle, split into the part common to all
inputs (norm of the mean error vector) and the per-input rest, bf16 and f16 operands."""
import json, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); warnings.simplefilter("ignore")
import numpy as np, torch
from outfitx_amd import synth, _lib as L
from oracle import np_oracle as O
from outfitx_amd.encoders import CLIPImageEncoder, CLIPTextEncoder
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
lib = L.load()
Wv, Wx = synth.vision_weights(7), synth.text_weights(7)
px = synth.pixel_values(11, n); ids, att = synth.token_batch(11, n, 64, synth.ragged_lengths(11, n, 3, 20))
ref_i = O.vit_forward(px, Wv); ref_t = O.text_forward(ids, att, Wx)
ie = CLIPImageEncoder(); ie.model.load_state_dict({k: torch.from_numpy(v) for k, v in Wv.items()}); ie = ie.cuda()
te = CLIPTextEncoder(); te.model.load_state_dict({k: torch.from_numpy(v) for k, v in Wx.items()}); te = te.cuda()
out = {}
for tp in ("bf16", "f16"):
    ie.tower_precision = tp; te.tower_precision = tp
    for fold in (2, 1, 0):
        lib.ofx_tune(6, fold)
        with torch.no_grad():
            gi = ie(torch.from_numpy(px).view(n, 1, 3, 224, 224).cuda(), normalize=False).view(n, -1).cpu().numpy()
            gt = te({"input_ids": torch.from_numpy(ids).view(n, 1, 64), "attention_mask": torch.from_numpy(att).view(n, 1, 64)}, normalize=False).view(n, -1).cpu().numpy()
        lib.ofx_tune(6, 2)
        for name, g, r in (("vit", gi, ref_i), ("text", gt, ref_t)):
            gn_, rn_ = g / np.linalg.norm(g, axis=1, keepdims=True), r / np.linalg.norm(r, axis=1, keepdims=True)
            dn = gn_ - rn_
            d = g - r
            scale = np.sqrt((r ** 2).mean())
            common = d.mean(0)
            out[f"{name}_{tp}_fold{fold}"] = {"rms_rel": round(float(np.sqrt((d ** 2).mean()) / scale), 5),
                                              "common_part_rms_rel": round(float(np.sqrt((common ** 2).mean()) / scale), 5),
                                              "per_input_part_rms_rel": round(float(np.sqrt(((d - common) ** 2).mean()) / scale), 5),
                                              "max_rel": round(float(np.abs(d).max() / np.abs(r).max()), 5),
                                              "after_l2norm_rms_rel": round(float(np.sqrt((dn ** 2).mean()) / np.sqrt((rn_ ** 2).mean())), 5),
                                              "after_l2norm_common_rms_rel": round(float(np.sqrt((dn.mean(0) ** 2).mean()) / np.sqrt((rn_ ** 2).mean())), 5)}
print(json.dumps(out, indent=1))
