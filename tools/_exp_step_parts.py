#!/usr/bin/env python3
"""Where the DP training step's time outside forward/backward goes (B = 256): weight re-pack, gradient accumulation into the
arena, clip, fused AdamW, arena zeroing."""
import os, sys, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outfitx_amd import synth
from outfitx_amd.trainer import FlatGrads
from src.models import OutfitX
from src.models.configs import ItemEncoderConfig, OutfitXConfig
cfg = OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip")); cfg.transformer.dropout = 0.0
m = OutfitX(cfg, train_precision="bf16")
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.outfit_transformer_weights(7).items()}, strict=False)
m = m.cuda().train()
params = [p for n, p in m.named_parameters() if not n.startswith("item_encoder.")]
fg = FlatGrads(params)
opt = torch.optim.AdamW(params, lr=2e-5, fused=True)
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def repack():
    with torch.no_grad():
        params[0].add_(0.0)             # bump a version counter (what optimizer.step does) -> the engine re-packs everything
    m._engine("bf16")
g = [torch.randn_like(p) for p in params]
res = {"ms_repack": t(repack), "ms_accumulate_75_grads": t(lambda: [p.grad.add_(x) for p, x in zip(params, g)]),
       "ms_clip": t(lambda: fg.clip_norm_(1.0)), "ms_fused_adamw": t(opt.step), "ms_zero_arena": t(fg.zero_)}
flat_p = fg.flatten_params_()
opt2 = torch.optim.AdamW([flat_p], lr=2e-5, fused=True)
res["ms_fused_adamw_flat"] = t(opt2.step)
opt3 = torch.optim.AdamW([flat_p], lr=2e-5, foreach=True)
res["ms_foreach_adamw_flat"] = t(opt3.step)
res["ms_repack_flat"] = t(lambda: (m.mark_weights_changed(), m._engine("bf16")))
print(json.dumps({k: round(v, 3) for k, v in res.items()}))
