#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running THE REFERENCE ITSELF (imported from /root/reference) on
seeded inputs.  Runs only in the build container (the reference never travels to the GPU box);
the fixtures it writes are data: expected outputs (+ checksums of the regenerated inputs).

    python oracle/gen_golden.py            # writes tests/golden/

How the reference is made importable here: transformers is imported first, then
empty stub modules stand in for open_clip / torchvision (neither is touched by the type='clip'
path); `from_pretrained` of the HF CLIP classes is patched to build the default-config
(ViT-B/32-shaped) models because no checkpoint is available offline; weights are then overwritten
by outfitx_amd.synth via load_state_dict(strict=True), so fixtures need no weights.
The HF image processor is configured as identity (pixel_values are fed pre-normalised) and the
tokenizer is replaced by a table lookup (the BPE vocab is not available offline).
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from outfitx_amd import synth  # noqa: E402

REF = os.environ.get("OUTFITX_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")
W_SEED = 7


def import_reference():
    import transformers
    from transformers import (CLIPImageProcessor, CLIPTextConfig, CLIPTextModelWithProjection,
                              CLIPTokenizer, CLIPVisionConfig, CLIPVisionModelWithProjection)

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m

    stub("open_clip")
    stub("torchvision")
    stub("torchvision.models", resnet18=None, ResNet18_Weights=None)
    stub("torchvision.transforms", transforms=None)
    stub("torchvision.transforms.transforms")
    stub("torchvision.transforms.v2")
    stub("torchvision.transforms.v2.functional", pad_video=None)

    CLIPVisionModelWithProjection.from_pretrained = classmethod(lambda cls, *a, **k: cls(CLIPVisionConfig()))
    CLIPTextModelWithProjection.from_pretrained = classmethod(lambda cls, *a, **k: cls(CLIPTextConfig()))

    class IdentityProcessor:
        """Stands in for CLIPImageProcessor: pixel_values are already resized+normalised."""
        size = {"shortest_edge": 224}

        def __call__(self, images=None, return_tensors="pt", **kw):
            from transformers import BatchFeature
            if not isinstance(images, torch.Tensor):
                images = torch.stack([torch.as_tensor(np.asarray(i)) for i in images])
            return BatchFeature({"pixel_values": images.float()})

    class TableTokenizer:
        """texts are keys '#<row>' into a registered id table."""
        table = None

        def __call__(self, text=None, **kw):
            rows = [int(t[1:]) for t in text]
            ids, att = TableTokenizer.table
            return {"input_ids": torch.as_tensor(ids[rows]), "attention_mask": torch.as_tensor(att[rows])}

    CLIPImageProcessor.from_pretrained = classmethod(lambda cls, *a, **k: IdentityProcessor())
    CLIPTokenizer.from_pretrained = classmethod(lambda cls, *a, **k: TableTokenizer())
    sys.path.insert(0, REF)
    import src.models as M
    import src.models.configs as C
    import src.models.datatypes as T
    from src.losses.focal_loss import FocalLoss
    from src.losses.set_wise_ranking_loss import SetWiseRankingLoss
    FocalLoss.SetWiseRankingLoss = SetWiseRankingLoss            # carried along without changing the tuple below
    from src.models.processor import OutfitXProcessorFactory
    return M, C, T, FocalLoss, OutfitXProcessorFactory, TableTokenizer, transformers.__version__


def t(a):
    return torch.as_tensor(np.ascontiguousarray(a))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    M, C, T, FocalLoss, Factory, TableTokenizer, tf_ver = import_reference()
    meta = dict(torch=torch.__version__, transformers=tf_ver, numpy=np.__version__, w_seed=W_SEED,
                note="reference code @2025-12-05 run with container library versions (pins: torch 2.5.0, transformers 4.48.3)")
    cfg = C.OutfitXConfig(item_encoder=C.ItemEncoderConfig(type="clip"))
    model = M.OutfitX(cfg).eval()
    sd = {k: t(v) for k, v in synth.full_state_dict(W_SEED).items()}
    missing = model.load_state_dict(sd, strict=True)
    print("loaded synthetic weights into the reference:", missing)
    CP, CIR = T.OutfitCompatibilityPredictionTask, T.OutfitComplementaryItemRetrievalTask

    def taps_of(layers):
        acc = []
        hooks = [l.register_forward_hook(lambda m, i, o: acc.append((o[0] if isinstance(o, tuple) else o)[:, 0].clone()))
                 for l in layers]
        return acc, hooks

    # ---- 1. outfit transformer, BASELINE config 1 shape (B=32, n=8 of L=16) ----------------
    def run_ot(tag, seed, B, n_items):
        emb, mask = synth.outfit_batch(seed, B, 16, n_items)
        txt = synth.unit_rows(seed, "target_text", B, 512)
        acc, hooks = taps_of(model.transformer_encoder.layers)
        cp = model(task=CP, outfit_embedding=t(emb), outfit_mask=t(mask)).numpy()
        cp_taps = torch.stack(acc).numpy(); acc.clear()
        cir = model(task=CIR, outfit_embedding=t(emb), outfit_mask=t(mask), target_item_text_embedding=t(txt)).numpy()
        cir_taps = torch.stack(acc).numpy()
        for h in hooks:
            h.remove()
        np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), seed=seed, B=B, n_items=np.asarray(n_items),
                            emb_crc=synth.checksum(emb), cp_logits=cp, cp_row0=cp_taps,
                            cir_emb=cir, cir_row0=cir_taps[-1], meta=str(meta))
        print(tag, cp[:4, 0], cir.shape)

    run_ot("ot_cfg1", 1235, 32, 8)
    n_rag = np.concatenate([np.arange(0, 17), synth.ragged_lengths(1236, 7, 1, 16)])
    run_ot("ot_ragged", 1236, len(n_rag), n_rag)

    # ---- 2. ViT-B/32 tower ---------------------------------------------------------------
    vis = model.item_encoder.image_enc.model
    px = synth.pixel_values(1237, 4)
    acc, hooks = taps_of(vis.vision_model.encoder.layers)
    img = vis(pixel_values=t(px)).image_embeds.numpy()
    cls = torch.stack(acc).numpy()
    for h in hooks:
        h.remove()
    np.savez_compressed(os.path.join(OUT, "vit_n4.npz"), seed=1237, N=4, px_crc=synth.checksum(px),
                        image_embeds=img, cls_per_layer=cls, meta=str(meta))
    print("vit", img[:, :3])

    # ---- 3. CLIP text tower ------------------------------------------------------------------
    txtm = model.item_encoder.text_enc.model
    n_real = np.array([2, 3, 8, 8, 17, 33, 64, 5])
    ids, att = synth.token_batch(1238, 8, 64, n_real)
    acc, hooks = taps_of(txtm.text_model.encoder.layers)
    te = txtm(input_ids=t(ids), attention_mask=t(att)).text_embeds.numpy()
    for h in hooks:
        h.remove()
    np.savez_compressed(os.path.join(OUT, "text_n8.npz"), seed=1238, n_real=n_real, ids_crc=synth.checksum(ids),
                        text_embeds=te, meta=str(meta))
    print("text", te[:, :3])

    # ---- 4. ItemEncoder.forward + _cp_forward(encoder_input_dict) + precompute_embeddings ----
    B, L = 2, 3
    px2 = synth.pixel_values(1239, B * L).reshape(B, L, 3, 224, 224)
    ids2, att2 = synth.token_batch(1239, B * L, 64, np.array([4, 8, 6, 3, 9, 12]))
    TableTokenizer.table = (ids2, att2)
    texts = [[f"#{b * L + l}" for l in range(L)] for b in range(B)]
    items = model.item_encoder(t(px2), texts).numpy()                       # [B,L,1024]
    mask2 = np.zeros((B, L), bool); mask2[1, 2] = True
    cp2 = model(task=CP, outfit_embedding=None, outfit_mask=t(mask2),
                encoder_input_dict={"images": t(px2), "texts": texts}).numpy()
    pe = model(task=T.OutfitPrecomputeEmbeddingTask, images=t(px2[:, :1]), texts=[[r[0]] for r in texts]).numpy()
    model.item_encoder.cfg.aggregation_method = "mean"
    items_mean = model.item_encoder(t(px2), texts).numpy()
    model.item_encoder.cfg.aggregation_method = "concat"
    np.savez_compressed(os.path.join(OUT, "item_encoder.npz"), seed=1239, item_emb=items, cp_logits=cp2, mask=mask2,
                        precomputed=pe, items_mean=items_mean, meta=str(meta))
    print("items", items.shape, "mean-agg shape", items_mean.shape, cp2.ravel())

    # ---- 5. FITB / CIR scoring ------------------------------------------------------------
    y = (synth.item_embeddings(1240, "y_hat", 64) * 3.0).astype(np.float32)
    cand = synth.item_embeddings(1240, "cand", 64, 4)
    d = torch.cdist(t(y).unsqueeze(1), t(cand), p=2).squeeze(1)
    fitb_idx = d.argmin(-1).numpy()
    Q = (synth.item_embeddings(1241, "queries", 100) * 3.0).astype(np.float32)
    P = synth.item_embeddings(1241, "pool", 5000)
    dm = torch.cdist(t(Q), t(P))
    tk = torch.topk(dm, k=50, largest=False)
    np.savez_compressed(os.path.join(OUT, "scoring.npz"), fitb_idx=fitb_idx, fitb_dist=d.numpy(),
                        topk_idx=tk.indices.numpy(), topk_dist=tk.values.numpy(), meta=str(meta))
    print("fitb", fitb_idx[:8], "topk", tk.indices[0, :5].tolist())

    # ---- 6. FocalLoss (next row N1) -------------------------------------------------------
    g = np.random.Generator(np.random.PCG64(1242))
    logits = (g.standard_normal(257) * 2).astype(np.float32)
    labels = (g.random(257) < 0.5).astype(np.float32)
    fl = float(FocalLoss(alpha=0.75, gamma=2.0)(t(logits), t(labels)))
    # ---- 7. CP collate processor (next row N3) --------------------------------------------
    proc = Factory.get_processor(CP, cfg)
    lens = [3, 8, 20, 1]
    rows = [synth.item_embeddings(1243, f"o{i}", n) for i, n in enumerate(lens)]
    batch = [(CP(outfit=[T.FashionItem(item_id=j, embedding=r[j]) for j in range(len(r))]), float(i % 2))
             for i, r in enumerate(rows)]
    bd = proc(batch)
    np.savez_compressed(os.path.join(OUT, "aux.npz"), focal_logits=logits, focal_labels=labels, focal_value=fl,
                        proc_lens=np.asarray(lens), proc_emb_crc=synth.checksum(bd["input_dict"]["outfit_embedding"].numpy()),
                        proc_mask=bd["input_dict"]["outfit_mask"].numpy(), proc_label=bd["label"].numpy(), meta=str(meta))
    print("focal", fl, "proc", tuple(bd["input_dict"]["outfit_embedding"].shape))

    # ---- 8. one CP training step (next row N1): reference model in train() mode, dropout 0, fp32 --------
    # cp_trainer:57-81 with accumulation 1 and no AMP: forward -> FocalLoss(.75, 2) -> backward -> clip_grad_norm_(1.0)
    # -> AdamW(lr) step.  Saved: loss, logits, per-parameter gradient norms, full small gradients, a strided sample of
    # the big ones, the clip norm and two post-step parameters.
    torch.set_grad_enabled(True)
    cfg0 = C.OutfitXConfig(item_encoder=C.ItemEncoderConfig(type="clip"))
    cfg0.transformer.dropout = 0.0
    m2 = M.OutfitX(cfg0)
    m2.load_state_dict(sd, strict=True)
    m2.train()
    n_tr = np.array([1, 2, 3, 5, 8, 8, 11, 16, 4, 7, 8, 6])
    emb, mask = synth.outfit_batch(1244, len(n_tr), 16, n_tr)
    labels = (np.arange(len(n_tr)) % 2).astype(np.float32)
    params = {k: v for k, v in m2.named_parameters() if not k.startswith("item_encoder.")}
    lr = 1e-3
    opt = torch.optim.AdamW(list(params.values()), lr=lr)
    opt.zero_grad()
    y_hat = m2(task=CP, outfit_embedding=t(emb), outfit_mask=t(mask)).squeeze(-1)
    loss = FocalLoss(alpha=0.75, gamma=2, reduction="mean")(y_hat=y_hat, y_true=t(labels))
    loss.backward()
    out = dict(seed=1244, n_items=n_tr, labels=labels, emb_crc=synth.checksum(emb), logits=y_hat.detach().numpy(),
               loss=float(loss), lr=lr, meta=str(meta))
    names, norms = [], []
    for k, v in params.items():
        if v.grad is None:
            continue
        g_ = v.grad.detach().numpy()
        names.append(k); norms.append(float(np.sqrt((g_.astype(np.float64) ** 2).sum())))
        if g_.size <= 4096:
            out["grad/" + k] = g_.copy()
        else:
            out["gsample/" + k] = g_.ravel()[::1009].copy()
    out["grad_names"] = np.asarray(names); out["grad_norms"] = np.asarray(norms)
    out["no_grad_names"] = np.asarray([k for k, v in params.items() if v.grad is None])
    out["clip_norm"] = float(torch.nn.utils.clip_grad_norm_(list(params.values()), max_norm=1.0))
    opt.step()
    out["post/outfit_token"] = m2.outfit_token.detach().numpy()
    out["post/cp_ffn.1.weight"] = m2.cp_ffn[1].weight.detach().numpy()
    out["post/transformer_encoder.layers.0.norm1.weight"] = m2.transformer_encoder.layers[0].norm1.weight.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "train_step.npz"), **out)
    print("train step: loss", float(loss), "clip norm", out["clip_norm"], "params with grad", len(names))

    # ---- 9. one CIR training step: reference model in train() mode (dropout 0, fp32), SetWiseRankingLoss(margin 2) ----
    # complementary_item_retrieval_trainer.py:73-88: y_hat = model(CIR batch); loss(batch_y, batch_y_hat, negatives, mask); backward.
    m2.load_state_dict(sd, strict=True)                      # §8's optimizer step moved the weights: back to the seeded ones
    m2.zero_grad(set_to_none=True)
    n_c = np.array([2, 5, 8, 1, 3, 7, 16, 4, 6, 9])
    Bc, Kn = len(n_c), 6
    emb_c, mask_c = synth.outfit_batch(1245, Bc, 16, n_c)
    txt_c = synth.unit_rows(1245, "target_text", Bc, 512)
    pos = synth.item_embeddings(1245, "pos", Bc)
    neg = synth.item_embeddings(1245, "neg", Bc * Kn).reshape(Bc, Kn, 1024)
    g_ = np.random.Generator(np.random.PCG64(1245))
    neg_mask = g_.random((Bc, Kn)) < 0.25
    neg_mask[:, 0] = False
    y_c = m2(task=CIR, outfit_embedding=t(emb_c), outfit_mask=t(mask_c), target_item_text_embedding=t(txt_c))
    loss_c = FocalLoss.SetWiseRankingLoss(margin=2.0)(batch_y=t(pos) * 3.0, batch_y_hat=y_c, batch_negative_samples=t(neg) * 3.0,
                                                     batch_negative_mask=t(neg_mask))
    loss_c.backward()
    outc = dict(seed=1245, n_items=n_c, K=Kn, neg_mask=neg_mask, emb_crc=synth.checksum(emb_c), y_hat=y_c.detach().numpy(), loss=float(loss_c),
                meta=str(meta))
    names, norms = [], []
    for k, v in params.items():
        if v.grad is None:
            continue
        gr = v.grad.detach().numpy()
        names.append(k); norms.append(float(np.sqrt((gr.astype(np.float64) ** 2).sum())))
        if gr.size <= 4096:
            outc["grad/" + k] = gr.copy()
        else:
            outc["gsample/" + k] = gr.ravel()[::1009].copy()
    outc["grad_names"] = np.asarray(names); outc["grad_norms"] = np.asarray(norms)
    outc["no_grad_names"] = np.asarray([k for k, v in params.items() if v.grad is None])
    np.savez_compressed(os.path.join(OUT, "train_step_cir.npz"), **outc)
    print("cir train step: loss", float(loss_c), "params with grad", len(names), "without", list(outc["no_grad_names"]))


if __name__ == "__main__":
    main()
