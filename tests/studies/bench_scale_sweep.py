#!/usr/bin/env python3
"""End-to-end parity at the HEADLINE batch size, per weight seed: the model runs BASELINE configs[1]'s 256 outfits x 8 items (so every
GEMM goes through the kernels the bench runs - the small test batches take the 128x128 split-K paths instead), the first 8 outfits
are the per-seed test batch and are compared with the fp32 restatement (oracle/torch_ref.py) exactly as bench.py's parity leg does.
    python tests/studies/bench_scale_sweep.py [seed ...]  > profiles/r02_seed_sweep_bench_scale.json
LADDER_SCHEMES=f16w2x,f16w2 selects the schemes (default: the default scheme only)."""
import json, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); warnings.simplefilter("ignore")
import numpy as np, torch
from outfitx_amd import synth, _lib as L
from oracle import torch_ref as TR
from src.models import OutfitX
from src.models.configs import ItemEncoderConfig, OutfitXConfig
from src.models.datatypes import OutfitCompatibilityPredictionTask as CP

seeds = [int(a) for a in sys.argv[1:]] or [7]
schemes = os.environ.get("LADDER_SCHEMES", L.DEFAULT_TOWER_PRECISION).split(",")
torch.set_num_threads(min(16, os.cpu_count() or 1))
dev = torch.device("cuda")
mean = torch.tensor(synth.CLIP_MEAN).view(1, 1, 3, 1, 1); std = torch.tensor(synth.CLIP_STD).view(1, 1, 3, 1, 1)
B, b, n = 256, 8, 8
gf = torch.Generator(device=dev); gf.manual_seed(4242)
fill = torch.randint(0, 256, (B - b, n, 3, 224, 224), generator=gf, device=dev, dtype=torch.uint8)
fill = ((fill.float() * (1 / 255.0) - mean.to(dev)) / std.to(dev)).contiguous()
fids, fatt = synth.token_batch(4242, (B - b) * n, 64, 8)
out = {s: {"errors": {}, "abs_errors": {}} for s in schemes}
for ws in seeds:
    gg = torch.Generator(); gg.manual_seed(9000 + ws)
    u8 = torch.randint(0, 256, (b, n, 3, 224, 224), generator=gg, dtype=torch.uint8)
    p8 = ((u8.float() * (1 / 255.0) - mean) / std).contiguous()
    ids, att = synth.token_batch(9000 + ws, b * n, 64, 8)
    V, T, O = TR.TorchRef(synth.vision_weights(ws)), TR.TorchRef(synth.text_weights(ws)), TR.TorchRef(synth.outfit_transformer_weights(ws))
    with torch.no_grad():
        ref = O.cp(TR.item_encoder(V, T, p8, torch.from_numpy(ids).view(b, n, 64), torch.from_numpy(att).view(b, n, 64)), torch.zeros(b, n, dtype=torch.bool)).numpy()
    px = torch.cat([p8.to(dev), fill], 0)
    tx = {"input_ids": torch.from_numpy(np.concatenate([ids, fids], 0)).view(B, n, 64), "attention_mask": torch.from_numpy(np.concatenate([att, fatt], 0)).view(B, n, 64)}
    sd = {k: torch.from_numpy(v) for k, v in synth.full_state_dict(ws).items()}
    for s in schemes:
        m = OutfitX(OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip")), tower_precision=s)
        m.load_state_dict(sd, strict=True); m = m.to(dev).eval()
        with torch.no_grad():
            got = m(task=CP, outfit_embedding=None, outfit_mask=torch.zeros(B, n, dtype=torch.bool, device=dev), encoder_input_dict={"images": px, "texts": tx})[:b].cpu().numpy()
        d = float(np.abs(got - ref).max())
        out[s]["errors"][str(ws)] = float(f"{d / np.abs(ref).max():.3g}"); out[s]["abs_errors"][str(ws)] = float(f"{d:.3g}")
        del m; torch.cuda.empty_cache()
    print(f"[bench-scale] seed {ws}: " + "  ".join(f"{s} {out[s]['errors'][str(ws)]:.2e}" for s in schemes), file=sys.stderr, flush=True)
for s in schemes:
    e = np.array(list(out[s]["errors"].values())); a = np.array(list(out[s]["abs_errors"].values()))
    out[s].update({"median_error": float(np.median(e)), "worst_error": float(e.max()), "seeds_over_1e-3": [k for k, v in out[s]["errors"].items() if v >= 1e-3],
                   "abs_error_median": float(np.median(a)), "abs_error_worst": float(a.max())})
print(json.dumps({"workload": "cfg2 at the headline batch size (256 outfits x 8 items through the bench's kernels); error of the first 8 outfits' CP logits vs the fp32 torch restatement, max|d| / max|ref| over those 8", "weight_seeds": seeds, "schemes": out}, indent=1))
