#!/usr/bin/env python3
"""How much of the default scheme's residual error is the ViT's attention core (q, k, v, P rounded once to f16)?  CPU emulation in
the manner of operand_scheme_cpu.py: ViT GEMMs = f16 activations x split weights (f16w2x) or split weights on patch / out-proj / fc2
only (f16w2), the attention core in f16, f16x2 (hi + lo) or exact; text tower and projection three-product, set transformer exact.
    python tests/studies/attn_core_cpu.py [n_seeds] [outfits]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from outfitx_amd import synth
from oracle.torch_ref import l2n
from operand_scheme_cpu import Net

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = 8
PROJ = {"visual_projection": ("f16x2", "f16x2")}
W2X = dict(PROJ); W2X["proj.weight|fc1|fc2|patch_embedding"] = ("f16", "f16x2")
W2 = dict(PROJ); W2["fc2|out_proj|patch_embedding"] = ("f16", "f16x2")
variants = {"f16w2x, core f16": (W2X, "f16"), "f16w2x, core f16x2": (W2X, "f16x2"), "f16w2x, core exact": (W2X, "f32"),
            "f16w2, core f16": (W2, "f16"), "f16w2, core exact": (W2, "f32")}
mean = torch.tensor(synth.CLIP_MEAN).view(1, 3, 1, 1); std = torch.tensor(synth.CLIP_STD).view(1, 3, 1, 1)
out = {v: [] for v in variants}
for ws in range(1, nseeds + 1):
    Wt, Wv, Wx = synth.outfit_transformer_weights(ws), synth.vision_weights(ws), synth.text_weights(ws)
    g = torch.Generator(); g.manual_seed(9000 + ws)
    u8 = torch.randint(0, 256, (k * n, 3, 224, 224), generator=g, dtype=torch.uint8)
    px = ((u8.float() * (1 / 255.0) - mean) / std).contiguous()
    ids, att = synth.token_batch(9000 + ws, k * n, 64, 8)
    ids, att = torch.from_numpy(ids[:, :8].copy()), torch.from_numpy(att[:, :8].copy())
    mask = torch.zeros(k, n, dtype=torch.bool)
    t0 = time.perf_counter()
    with torch.no_grad():
        S = Net(Wt, "f32", "f32")
        rv, rt = Net(Wv, "f32", "f32").vit(px), Net(Wx, "f32", "f32").text(ids, att)
        ref = S.cp(torch.cat([l2n(rv), l2n(rt)], -1).view(k, n, -1), mask)
        gt = Net(Wx, "f16x2", "f16x2", attn_mode="f32").text(ids, att)
        for name, (sites, core) in variants.items():
            gv = Net(Wv, "f16", "f16", attn_mode=core, sites=sites).vit(px)
            got = S.cp(torch.cat([l2n(gv), l2n(gt)], -1).view(k, n, -1), mask)
            out[name].append(float((got - ref).abs().max() / ref.abs().max()))
    print(f"seed {ws}: {time.perf_counter() - t0:.0f} s  " + "  ".join(f"[{v}] {out[v][-1]:.2e}" for v in variants), file=sys.stderr, flush=True)
print(json.dumps({"metric": "max|d| / max|ref| over the batch", "outfits": k, "weight_seeds": nseeds,
                  "variants": {v: {"median": float(np.median(e)), "max": max(e), "all": [float(f"{x:.3g}") for x in e]} for v, e in out.items()}}, indent=1))
