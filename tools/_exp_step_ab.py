import sys, time, warnings, ctypes as C
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))); warnings.simplefilter('ignore')
import torch, numpy as np
from outfitx_amd import synth, _lib as L
from src.models import OutfitX
from src.models.configs import ItemEncoderConfig, OutfitXConfig
from src.models.datatypes import OutfitCompatibilityPredictionTask as CP
dev = torch.device('cuda', 0)
model = OutfitX(OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip")))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.full_state_dict(7).items()}, strict=True)
model = model.to(dev).eval()
B, n = 256, 8
px = torch.randn(B, n, 3, 224, 224, device=dev)
ids_np, att_np = synth.token_batch(1, B * n, 64, 8)
texts = {"input_ids": torch.from_numpy(ids_np).view(B, n, 64).pin_memory(), "attention_mask": torch.from_numpy(att_np).view(B, n, 64).pin_memory()}
mask = torch.zeros(B, n, dtype=torch.bool, device=dev)
def step():
    with torch.no_grad():
        return model(task=CP, outfit_embedding=None, outfit_mask=mask, encoder_input_dict={"images": px, "texts": texts})
lib = L.load()
for _ in range(3): step()
torch.cuda.synchronize()
# host time to ISSUE one step (no sync inside the call if this is a few ms) vs the GPU time of the step
iss = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    iss.append((t1 - t0, t2 - t0))
print("issue ms / complete ms per step:", " ".join(f"{a*1e3:.2f}/{b*1e3:.2f}" for a, b in iss), flush=True)
for rnd in range(3):
    for knob, val in [tuple(int(x) for x in a.split("=")) for a in sys.argv[1:]] or [(8, 1), (8, 0)]:
        if knob == 100: model.tower_fed_precision = "f16" if val else None      # set transformer fed by in-call towers: f16 vs bf16x3
        else: lib.ofx_tune(knob, val)
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6): step()
        torch.cuda.synchronize()
        print(f"knob {knob} = {val}: {(time.perf_counter()-t0)/6*1e3:7.2f} ms/step", flush=True)
