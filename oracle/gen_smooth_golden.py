#!/usr/bin/env python3
"""Photo-like inputs through the reference (test infrastructure, build container only): tests/golden/cfg2_smooth_images.npz.

Every other end-to-end fixture feeds uniform-noise images.  This one runs the reference's _cp_forward(encoder_input_dict) (outfit_x.py:120-144) on 64 outfits x 8
items of outfitx_amd.synth.smooth_pixel_values - smooth backgrounds, flat rectangles, neighbouring-pixel correlation 0.99: patches that are nearly constant, so
the patch embedding and the first LayerNorms see inputs dominated by their common-mode component - for weight seeds 7 and 21, and stores the logits
(+ checksums of the regenerated inputs).        python oracle/gen_smooth_golden.py
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from outfitx_amd import synth  # noqa: E402
import gen_golden as G  # noqa: E402

IN_SEED, B, N_ITEMS, CHUNK = 1251, 64, 8, 16


def main():
    torch.set_grad_enabled(False)
    torch.set_num_threads(int(os.environ.get("OFX_GEN_THREADS", "8")))
    M, C, T, _, _, TableTokenizer, tf_ver = G.import_reference()
    px = synth.smooth_pixel_values(IN_SEED, B * N_ITEMS).reshape(B, N_ITEMS, 3, 224, 224)
    ids, att = synth.token_batch(IN_SEED, B * N_ITEMS, 64, 8)
    TableTokenizer.table = (ids, att)
    model = M.OutfitX(C.OutfitXConfig(item_encoder=C.ItemEncoderConfig(type="clip"))).eval()
    out = dict(in_seed=IN_SEED, outfits=B, items=N_ITEMS, px_crc=synth.checksum(px[:2]), ids_crc=synth.checksum(ids),
               meta=f"reference OutfitX._cp_forward(encoder_input_dict) fp32 CPU on synth.smooth_pixel_values; torch {torch.__version__}, transformers {tf_ver}")
    for ws in (sys.argv[1:] or ["7", "21"]):
        t0 = time.time()
        model.load_state_dict({k: G.t(v) for k, v in synth.variant_state_dict(ws).items()}, strict=True)
        rows = []
        for b0 in range(0, B, CHUNK):
            texts = [[f"#{(b0 + b) * N_ITEMS + l}" for l in range(N_ITEMS)] for b in range(CHUNK)]
            y = model(task=T.OutfitCompatibilityPredictionTask, outfit_embedding=None, outfit_mask=torch.zeros(CHUNK, N_ITEMS, dtype=torch.bool),
                      encoder_input_dict={"images": G.t(px[b0:b0 + CHUNK]), "texts": texts})
            rows.append(y.reshape(-1).numpy().astype(np.float32))
        out[f"w{ws}"] = np.concatenate(rows)
        print(f"seed {ws}: max|logit| {np.abs(out[f'w{ws}']).max():.4f}  ({time.time() - t0:.0f} s)", flush=True)
    np.savez_compressed(os.path.join(G.OUT, "cfg2_smooth_images.npz"), **out)


if __name__ == "__main__":
    main()
