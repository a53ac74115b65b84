#!/usr/bin/env python3
"""The towers' precision / throughput ladder in one run: for every `tower_precision` scheme the cfg2 step time
(256 outfits x 8 items, 20 steps after 5 warm-up, text tower on the side stream) and the end-to-end CP-logit error against the fp32
oracle on a list of weight seeds (8 outfits each).   python tests/studies/precision_ladder.py [seed ...]  > profiles/r02_precision_ladder.json
LADDER_SCHEMES=f16w2,f16w2x LADDER_TIMING=0 restricts the schemes / skips the throughput leg (the 40-seed sweep of profiles/r02_seed_sweep_gpu.json)."""
import json, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); warnings.simplefilter("ignore")
import numpy as np, torch
from outfitx_amd import synth
from oracle import np_oracle as O
from src.models import OutfitX
from src.models.configs import ItemEncoderConfig, OutfitXConfig
from src.models.datatypes import OutfitCompatibilityPredictionTask as CP

seeds = [int(a) for a in sys.argv[1:]] or [4, 6, 14]
schemes = os.environ.get("LADDER_SCHEMES", "bf16,f16,f16w2,f16w2x,f16x3").split(",")
timing = os.environ.get("LADDER_TIMING", "1") != "0"
dev = torch.device("cuda")
mean = torch.tensor(synth.CLIP_MEAN).view(1, 1, 3, 1, 1); std = torch.tensor(synth.CLIP_STD).view(1, 1, 3, 1, 1)
out = {s: {"errors": {}} for s in schemes}

# ---- throughput (weight seed 7, the bench's batch shape)
B, n = 256, 8
g = torch.Generator(device=dev); g.manual_seed(1236)
u8 = torch.randint(0, 256, (B, n, 3, 224, 224), generator=g, device=dev, dtype=torch.uint8)
px = ((u8.float() * (1 / 255.0) - mean.to(dev)) / std.to(dev)).contiguous(); del u8
ids, att = synth.token_batch(1236, B * n, 64, 8)
texts = {"input_ids": torch.from_numpy(ids).view(B, n, 64).pin_memory(), "attention_mask": torch.from_numpy(att).view(B, n, 64).pin_memory()}
mask = torch.zeros(B, n, dtype=torch.bool, device=dev)
sd7 = {k: torch.from_numpy(v) for k, v in synth.full_state_dict(7).items()}
for s in (schemes if timing else []):
    m = OutfitX(OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip")), tower_precision=s)
    m.load_state_dict(sd7, strict=True); m = m.to(dev).eval()
    with torch.no_grad():
        for _ in range(5):
            m(task=CP, outfit_embedding=None, outfit_mask=mask, encoder_input_dict={"images": px, "texts": texts})
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            m(task=CP, outfit_embedding=None, outfit_mask=mask, encoder_input_dict={"images": px, "texts": texts})
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    out[s]["ms_per_step"] = round(dt * 1e3, 2); out[s]["outfits_per_s"] = round(B / dt, 1)
    print(f"[ladder] {s}: {dt * 1e3:.2f} ms/step", file=sys.stderr, flush=True)
    del m; torch.cuda.empty_cache()
del px

# ---- end-to-end error per weight seed
b = 8
for ws in seeds:
    gg = torch.Generator(); gg.manual_seed(9000 + ws)
    u8 = torch.randint(0, 256, (b, n, 3, 224, 224), generator=gg, dtype=torch.uint8)
    p = ((u8.float() * (1 / 255.0) - mean) / std).contiguous()
    ids, att = synth.token_batch(9000 + ws, b * n, 64, 8)
    tx = {"input_ids": torch.from_numpy(ids).view(b, n, 64), "attention_mask": torch.from_numpy(att).view(b, n, 64)}
    mk = np.zeros((b, n), bool)
    ref = O.cp_forward(O.item_encoder(p.numpy(), ids.reshape(b, n, 64), att.reshape(b, n, 64), synth.vision_weights(ws), synth.text_weights(ws)), mk,
                       synth.outfit_transformer_weights(ws))
    sd = {k: torch.from_numpy(v) for k, v in synth.full_state_dict(ws).items()}
    for s in schemes:
        m = OutfitX(OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip")), tower_precision=s)
        m.load_state_dict(sd, strict=True); m = m.to(dev).eval()
        with torch.no_grad():
            got = m(task=CP, outfit_embedding=None, outfit_mask=torch.from_numpy(mk).to(dev), encoder_input_dict={"images": p.to(dev), "texts": tx}).cpu().numpy()
        out[s]["errors"][str(ws)] = float(f"{np.abs(got - ref).max() / np.abs(ref).max():.3g}")
        del m; torch.cuda.empty_cache()
    print(f"[ladder] seed {ws}: " + "  ".join(f"{s} {out[s]['errors'][str(ws)]:.1e}" for s in schemes), file=sys.stderr, flush=True)
for s in schemes:
    e = np.array(sorted(out[s]["errors"].values()))
    out[s]["worst_error"] = float(e[-1]); out[s]["median_error"] = float(np.median(e)); out[s]["seeds_over_1e-3"] = int((e >= 1e-3).sum())
print(json.dumps({"workload": "cfg2: 256 outfits x 8 items per step (throughput, weight seed 7); 8 outfits x 8 items per weight seed (error vs the fp32 numpy oracle, max|d| / max|ref|)",
                  "weight_seeds": seeds, "schemes": out}, indent=1))
