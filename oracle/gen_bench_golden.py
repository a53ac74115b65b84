#!/usr/bin/env python3
"""CP logits of bench.py's WHOLE batch (BASELINE configs[1]: 256 outfits x 8 items, input seed 1236) computed by THE REFERENCE
ITSELF (imported from /root/reference exactly as oracle/gen_golden.py does: src.models.OutfitX._cp_forward with
encoder_input_dict, outfit_x.py:120-144) on the CPU in fp32, one row per weight seed.  Runs only in the build container; the
fixture it writes is data (expected outputs + checksums of the regenerated inputs):

    python oracle/gen_bench_golden.py 7 44 89 97 99        # -> tests/golden/cfg2_bench_logits.npz  (resumable: seeds are appended)
    python oracle/gen_bench_golden.py 3o1 3o2              # weight seed 3 with synth.outlier_channels level 1 / 2 (rows "w3o1", "w3o2")
    python oracle/gen_bench_golden.py 5t3 11t3             # weight seeds 5 / 11 with heavy-tailed (Student-t, 3 degrees of freedom) matrices (synth.heavy_tailed)

bench.py's parity leg, tests/test_gpu_model.py::test_cfg2_bench_batch_* and tests/studies/bench_scale_sweep.py compare all 256
logits of the HIP path with these rows (max|d| / max|ref| over the batch, the north star's metric).  ~2 CPU-minutes per seed on
8 cores, which is why it is a fixture and not computed on the GPU box.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from outfitx_amd import synth  # noqa: E402
from oracle.gen_golden import import_reference, t  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "cfg2_bench_logits.npz")
IN_SEED, B, N_ITEMS, CHUNK = 1236, 256, 8, 16


def main():
    seeds = sys.argv[1:] or ["7"]          # "<seed>" or "<seed>o<level>": that seed's weights with massive ViT channels (synth.outlier_channels)
    torch.set_grad_enabled(False)
    torch.set_num_threads(int(os.environ.get("OFX_GEN_THREADS", "6")))
    M, C, T, _, _, TableTokenizer, tf_ver = import_reference()
    CP = T.OutfitCompatibilityPredictionTask
    px, ids, att = synth.bench_batch(IN_SEED, B, N_ITEMS)
    have = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    have.update(in_seed=IN_SEED, outfits=B, items=N_ITEMS, px_crc=synth.checksum(px[:2]), ids_crc=synth.checksum(ids),
                meta=f"reference OutfitX._cp_forward(encoder_input_dict) fp32 CPU; torch {torch.__version__}, transformers {tf_ver}")
    TableTokenizer.table = (ids, att)
    model = M.OutfitX(C.OutfitXConfig(item_encoder=C.ItemEncoderConfig(type="clip"))).eval()
    for ws in seeds:
        if f"w{ws}" in have:
            continue
        t0 = time.time()
        sd = synth.variant_state_dict(ws)
        model.load_state_dict({k: t(v) for k, v in sd.items()}, strict=True)
        rows = []
        for b0 in range(0, B, CHUNK):
            texts = [[f"#{(b0 + b) * N_ITEMS + l}" for l in range(N_ITEMS)] for b in range(CHUNK)]
            y = model(task=CP, outfit_embedding=None, outfit_mask=torch.zeros(CHUNK, N_ITEMS, dtype=torch.bool),
                      encoder_input_dict={"images": t(px[b0:b0 + CHUNK]), "texts": texts})
            rows.append(y.reshape(-1).numpy().astype(np.float32))
        have[f"w{ws}"] = np.concatenate(rows)
        np.savez_compressed(OUT + ".tmp.npz", **have)
        os.replace(OUT + ".tmp.npz", OUT)
        print(f"seed {ws}: max|logit| {np.abs(have[f'w{ws}']).max():.4f}  ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
