#!/usr/bin/env python3
"""STRING inputs through the reference (test infrastructure, build container only): tests/golden/text_strings.npz.

The reference's text path starts from Python strings (text_encoders/clip_text_encoder.py:42-50: CLIPTokenizer(text, max_length=64,
padding='max_length', truncation=True) -> CLIPTextModelWithProjection).  The fashion-clip vocabulary is not available offline, but
the tokenizer CLASS is: this script gives the reference's own CLIPTokenizer a synthetic vocabulary of CLIP's format
(outfitx_amd.synth.write_clip_vocabulary, regenerated bit for bit on the GPU box), runs the reference on lists of strings
(BaseTextEncoder.forward, ItemEncoder.forward, _cp_forward(encoder_input_dict), precompute_embeddings) with the seeded weights, and
stores the strings, the token ids the reference's tokenizer produced and the outputs.

    python oracle/gen_string_golden.py
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from outfitx_amd import synth  # noqa: E402
import gen_golden as G  # noqa: E402

STRINGS = [["Red leather bag", "wool coat & hat"],
           ["blue jeans, top", "SHOES!"],
           ["", "a skirt made of red wool with a blue leather hat and a top " * 6]]          # empty text; a text truncated at 64 tokens


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    from transformers import CLIPTokenizer
    real = CLIPTokenizer.from_pretrained.__func__
    vocab_dir = synth.write_clip_vocabulary(os.path.join(tempfile.mkdtemp(), "clip_synth"))
    M, C, T, _, _, _, tf_ver = G.import_reference()
    CLIPTokenizer.from_pretrained = classmethod(lambda cls, *a, **k: real(cls, vocab_dir))       # the REAL tokenizer, on the synthetic vocabulary
    cfg = C.OutfitXConfig(item_encoder=C.ItemEncoderConfig(type="clip"))
    model = M.OutfitX(cfg).eval()
    model.load_state_dict({k: G.t(v) for k, v in synth.full_state_dict(G.W_SEED).items()}, strict=True)
    enc = model.item_encoder.text_enc
    flat = [s for row in STRINGS for s in row]
    tok = enc.tokenizer(text=flat, max_length=64, padding="max_length", truncation=True, return_tensors="pt")
    B, L = len(STRINGS), len(STRINGS[0])
    px = synth.pixel_values(1250, B * L).reshape(B, L, 3, 224, 224)
    raw = enc(STRINGS, normalize=False).numpy()                                                  # [B, L, 512]
    items = model.item_encoder(G.t(px), STRINGS).numpy()
    mask = np.zeros((B, L), bool); mask[2, 1] = True
    cp = model(task=T.OutfitCompatibilityPredictionTask, outfit_embedding=None, outfit_mask=G.t(mask),
               encoder_input_dict={"images": G.t(px), "texts": STRINGS}).numpy()
    pe = model(task=T.OutfitPrecomputeEmbeddingTask, images=G.t(px[:, :1]), texts=[[r[0]] for r in STRINGS]).numpy()
    out = os.path.join(G.OUT, "text_strings.npz")
    np.savez_compressed(out, strings=np.asarray(flat), rows=B, cols=L, input_ids=tok["input_ids"].numpy(), attention_mask=tok["attention_mask"].numpy(),
                        text_embeds=raw, item_emb=items, mask=mask, cp_logits=cp, precomputed=pe, px_seed=1250, w_seed=G.W_SEED,
                        meta=str(dict(torch=torch.__version__, transformers=tf_ver, tokenizer=type(enc.tokenizer).__name__)))
    print("tokens per string:", tok["attention_mask"].sum(-1).tolist(), "| cp", cp.ravel(), "->", out)


if __name__ == "__main__":
    main()
