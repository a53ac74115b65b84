#!/usr/bin/env python3
"""Single-GPU timings of the other BASELINE.json configs (parity-test cases, not bench.py lines):
  cfg1  CP forward, 32 precomputed-embedding outfits (8 of 16 items)       [+ hipGraph replay]
  cfg3  FITB: CIR forward + 4-candidate argmin, 1024 outfits
  cfg4  CIR: 1000 queries vs 100k-item pool, k=50 (unsharded, and one 12.5k shard = 1/8 of the 8-GPU layout)
Prints one JSON line per config.   python tools/bench_configs.py [cfg1] [cfg1x] [cfg3] [cfg4]   (default: all; cfg1 = 32 outfits only, cfg1x = its 256 / 1024-outfit variants; rocprofv3 passes name one)"""
import json, os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter("ignore")
from outfitx_amd import synth
from outfitx_amd.engine import Engine, fitb_argmin
from src.models import OutfitX
from src.models.configs import ItemEncoderConfig, OutfitXConfig
from src.models.datatypes import OutfitCompatibilityPredictionTask as CP, OutfitComplementaryItemRetrievalTask as CIR

dev = torch.device("cuda", 0)
# (OFX_TUNE=17:0 in the environment sets a knob at library load, outfitx_amd/_lib.py: e.g. top-k through the distance matrix + radix select)
model = OutfitX(OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip")))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.outfit_transformer_weights(7).items()}, strict=False)
model = model.to(dev).eval()
model.precision = os.environ.get("OFX_OT_PRECISION", "bf16x3")        # outfit-transformer operand scheme: bf16x3 (default) | f16w2 | f16 | bf16
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


only = set(a for a in sys.argv[1:] if a.startswith("cfg"))
with torch.no_grad():
    for B in ((32, 256, 1024) if not only else ((32,) if "cfg1" in only else ()) + ((256, 1024) if "cfg1x" in only else ())):
        emb, mask = synth.outfit_batch(1235, B, 16, 8)
        e, m = cu(emb), cu(mask)
        f = lambda: model(task=CP, outfit_embedding=e, outfit_mask=m)
        t = timeit(f)
        rec = {"config": "cfg1" if B == 32 else f"cfg1-B{B}", "what": f"CP forward, {B} precomputed outfits (8 of 16 items), {model.precision}", "ms": round(t * 1e3, 4), "outfits_per_s": round(B / t, 1)}
        if B == 32:       # cfg1 is weight-stream bound - 51,155,313 weights x 2 B once per forward (x3 operand copies in bf16x3)
            rec["weight_stream_GBps_bf16"] = round(51_155_313 * 2 / t / 1e9, 1)
            rec["weight_stream_GBps_as_executed_x3"] = round(51_155_313 * 6 / t / 1e9, 1)
        # the same launch sequence captured in a hipGraph (launch-bound at small B)
        out = f(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            f()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            out_g = f()
        tg = timeit(g.replay)
        rec.update({"graph_ms": round(tg * 1e3, 4), "graph_outfits_per_s": round(B / tg, 1), "graph_equal": bool(torch.equal(out, out_g))})
        print(json.dumps(rec), flush=True)

    if only and "cfg3" not in only and "cfg4" not in only:
        sys.exit(0)
    B = 1024
    emb, mask = synth.outfit_batch(51, B, 16, 8)
    txt = synth.unit_rows(51, "t", B, 512); cand = synth.item_embeddings(51, "cand", B, 4)
    e, m, tx, cd = cu(emb), cu(mask), cu(txt), cu(cand)
    def fitb():
        y = model(task=CIR, outfit_embedding=e, outfit_mask=m, target_item_text_embedding=tx)
        return fitb_argmin(y, cd)
    t = timeit(fitb, 30) if (not only or "cfg3" in only) else float("nan")
    if not only or "cfg3" in only:
        print(json.dumps({"config": "cfg3", "what": "FITB: CIR forward + cdist/argmin over 4 candidates, 1024 outfits", "ms": round(t * 1e3, 3), "outfits_per_s": round(B / t, 1)}), flush=True)
    if only and "cfg4" not in only:
        sys.exit(0)

    nq, npool, k = 1000, 100_000, 50
    emb, mask = synth.outfit_batch(61, nq, 16, 8)
    txt = synth.unit_rows(61, "t", nq, 512)
    P = cu(synth.item_embeddings(61, "pool", npool))
    e, m, tx = cu(emb), cu(mask), cu(txt)
    eng = model._engine()
    q = model(task=CIR, outfit_embedding=e, outfit_mask=m, target_item_text_embedding=tx)
    t_fwd = timeit(lambda: model(task=CIR, outfit_embedding=e, outfit_mask=m, target_item_text_embedding=tx), 30)
    t_full = timeit(lambda: eng.l2_topk(q, P, k), 10, 2)
    t_shard = timeit(lambda: eng.l2_topk(q, P[:12_500], k), 20, 2)
    print(json.dumps({"config": "cfg4", "what": "CIR: 1000 queries vs 100k pool, top-50, fp32-exact", "query_forward_ms": round(t_fwd * 1e3, 3),
                      "topk_unsharded_ms": round(t_full * 1e3, 3), "topk_one_of_8_shards_ms": round(t_shard * 1e3, 3),
                      "distance_gemm_tflops_fp32": round(2 * nq * npool * 1024 / t_full / 1e12, 1)}), flush=True)
