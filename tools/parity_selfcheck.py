#!/usr/bin/env python3
"""Parity self-check on YOUR weights, without the CPU reference: the CP logits of the default operand scheme against the near-fp32 three-product scheme
(tower_precision='f16x3', precision='bf16x3': every GEMM in three-product arithmetic, fp32 attention; ~1e-4 from the fp32 reference on well-conditioned
weights, 1.8x the time) on one batch.  Their disagreement, max|d| / max|ref| over the batch, estimates the default's distance from fp32; when it approaches 1e-3 -
peaked attention, DESIGN.md section 2 - run the slower scheme, or validate against the reference itself.
    python tools/parity_selfcheck.py [--checkpoint model.pth] [--weights KEY] [--outfits 256] [--items 8] [--scheme f16w2x]
--checkpoint: a reference checkpoint ({'model': state_dict} or a bare state_dict, strict load); otherwise synthetic weights by fixture key (--weights 7, 9s1.5, 5t3 ...).
Inputs: bench.py's synthetic batch (the images / token ids themselves matter little: the disagreement is a property of the weights)."""
import argparse, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); warnings.simplefilter("ignore")
import numpy as np, torch
from outfitx_amd import synth, _lib as L
from src.models import OutfitX
from src.models.configs import ItemEncoderConfig, OutfitXConfig
from src.models.datatypes import OutfitCompatibilityPredictionTask as CP


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint"); ap.add_argument("--weights", default="7"); ap.add_argument("--outfits", type=int, default=256)
    ap.add_argument("--items", type=int, default=8); ap.add_argument("--scheme", default=L.DEFAULT_TOWER_PRECISION)
    ap.add_argument("--f32-attention", type=int, default=1, help="1 (default): the three-product run also keeps the ViT's q | k | v in fp32 and runs fp32 attention (ofx_tune(20, 1): slow, "
                    "and 2.5x closer to the reference under peaked attention than the MFMA attention on f16 q | k | v that every timed scheme uses)")
    a = ap.parse_args()
    if a.checkpoint:
        sd = torch.load(a.checkpoint, map_location="cpu")
        sd = sd.get("model", sd)
    else:
        sd = {k: torch.from_numpy(v) for k, v in synth.variant_state_dict(a.weights).items()}
    px, ids, att = synth.bench_batch(1236, a.outfits, a.items)
    px = torch.from_numpy(px).cuda()
    tx = {"input_ids": torch.from_numpy(ids).view(a.outfits, a.items, 64), "attention_mask": torch.from_numpy(att).view(a.outfits, a.items, 64)}
    mask = torch.zeros(a.outfits, a.items, dtype=torch.bool, device="cuda")
    out = {}
    for s in (a.scheme, "f16x3"):
        L.check(L.load().ofx_tune(20, 1 if (s == "f16x3" and a.f32_attention) else 0))
        m = OutfitX(OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip")), tower_precision=s)
        m.load_state_dict(sd, strict=True); m = m.cuda().eval()
        with torch.no_grad():
            out[s] = m(task=CP, outfit_embedding=None, outfit_mask=mask, encoder_input_dict={"images": px, "texts": tx}).float().cpu().numpy().reshape(-1).astype(np.float64)
        del m; torch.cuda.empty_cache()
    L.check(L.load().ofx_tune(20, 0))
    d = float(np.abs(out[a.scheme] - out["f16x3"]).max()); r = float(np.abs(out["f16x3"]).max())
    print(f"{a.scheme} vs f16x3 on {a.outfits} outfits x {a.items} items: max|d| {d:.3e}, max|ref| {r:.3f}, max|d| / max|ref| = {d / r:.2e}  (north star's bound vs fp32: 1e-3)")


if __name__ == "__main__":
    main()
