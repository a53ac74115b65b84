#!/usr/bin/env python3
"""N3 measurement: batch hand-over to the CP forward, padded tensors vs indices into a device-resident table.

  padded : processor pads fp32 [B,16,1024] on the host (outfitx_amd.processor, the reference's collate output) ->
           pinned H2D copy -> forward
  indexed: OutfitXIndexedProcessor emits int32 indices + offsets -> H2D of a few KB -> forward gathers rows in HBM

    python tools/bench_collate.py [--batch 3072] [--items 8]
"""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outfitx_amd import synth
from outfitx_amd.embedding_store import EmbeddingTable


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=3072)      # the reference's CP batch size (base_train_config.py)
    ap.add_argument("--items", type=int, default=8)
    ap.add_argument("--table", type=int, default=100_000)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    from src.models import OutfitX
    from src.models.configs import ItemEncoderConfig, OutfitXConfig
    from src.models.datatypes import FashionItem, OutfitCompatibilityPredictionTask as CP
    from src.models.processor import OutfitXProcessorFactory
    from outfitx_amd.processor import OutfitXIndexedProcessor
    cfg = OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip"))
    m = OutfitX(cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.outfit_transformer_weights(7).items()}, strict=False)
    m = m.cuda().eval()
    g = np.random.default_rng(0)
    emb = synth.item_embeddings(5, "table", a.table)
    table = EmbeddingTable(np.arange(a.table), emb)
    m.set_embedding_table(torch.from_numpy(emb))
    n = g.integers(max(1, a.items - 4), a.items + 5, a.batch).clip(1, 16)
    ids = [g.integers(0, a.table, k) for k in n]
    dense_batch = [(CP(outfit=[FashionItem(item_id=int(i), embedding=emb[i]) for i in r]), 0.0) for r in ids]
    index_batch = [(CP(outfit=[FashionItem(item_id=int(i)) for i in r]), 0.0) for r in ids]
    pd = OutfitXProcessorFactory.get_processor(CP, cfg)
    pi = OutfitXIndexedProcessor(CP, cfg, id_to_row=table.index())

    def t(fn):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps): out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.reps * 1e3, out

    res = {"batch": a.batch, "mean_items": float(n.mean())}
    res["ms_collate_padded"], bd = t(lambda: pd(dense_batch))
    res["ms_collate_indexed"], bi = t(lambda: pi(index_batch))
    e, k = bd["input_dict"]["outfit_embedding"].pin_memory(), bd["input_dict"]["outfit_mask"].pin_memory()
    res["h2d_bytes_padded"] = e.numel() * 4 + k.numel()
    res["h2d_bytes_indexed"] = bi["input_dict"]["item_index"].numel() * 4 + bi["input_dict"]["cu_seqlens"].numel() * 4
    with torch.no_grad():
        res["ms_h2d_forward_padded"], ya = t(lambda: m(task=CP, outfit_embedding=e.cuda(non_blocking=True), outfit_mask=k.cuda(non_blocking=True)))
        ii, cc = bi["input_dict"]["item_index"].pin_memory(), bi["input_dict"]["cu_seqlens"].pin_memory()
        res["ms_h2d_forward_indexed"], yb = t(lambda: m(task=CP, item_index=ii, cu_seqlens=cc))
        ed, kd = e.cuda(), k.cuda()
        res["ms_forward_resident_padded"], _ = t(lambda: m(task=CP, outfit_embedding=ed, outfit_mask=kd))
    res["identical"] = bool(torch.equal(ya, yb))
    print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in res.items()}))


if __name__ == "__main__":
    main()
