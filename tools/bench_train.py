#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: the CP trainer step on precomputed embeddings (cp_trainer:57-81) —
forward -> FocalLoss(.75, 2) -> backward -> clip_grad_norm_(1.0) -> AdamW step — through src.models.OutfitX in train() mode
(HIP tape forward + hand-written backward).  Per-GPU batch 256 (= 2048 / 8 ranks) by default.

    python tools/bench_train.py [--batch 256] [--items 8] [--steps 20] [--precision bf16] [--eager]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_train.py   (DP, RCCL)

ms_step is the reference-shaped step (torch clip_grad_norm_ + default AdamW on per-tensor grads); ms_step_dp is the
outfitx_amd.trainer.CPTrainer step (flat gradient arena, one all-reduce, fused AdamW), accumulation 1 = an optimizer step
and an all-reduce on EVERY micro-batch (the worst case; the reference default accumulates 4).

--eager also times the same step written with plain torch modules (nn.TransformerEncoder under bf16 autocast, what the
reference's trainer executes) on the same GPU, for a like-for-like ratio.  Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outfitx_amd import synth  # noqa: E402


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--items", type=int, default=8)
    ap.add_argument("--pad", type=int, default=16)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--polyvore", action="store_true", help="SURVEY.md §8d config 5 inputs: outfit lengths uniform{2..8} padded to 16, labels Bernoulli(0.5)")
    a = ap.parse_args()
    from src.losses import FocalLoss
    from src.models import OutfitX
    from src.models.configs import ItemEncoderConfig, OutfitXConfig
    from src.models.datatypes import OutfitCompatibilityPredictionTask as CP
    cfg = OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip"))
    cfg.transformer.dropout = 0.0
    m = OutfitX(cfg, train_precision=a.precision)
    sd = synth.outfit_transformer_weights(7)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    m = m.cuda().train()
    params = [v for k, v in m.named_parameters() if not k.startswith("item_encoder.")]
    opt = torch.optim.AdamW(params, lr=2e-5)
    n_items = synth.ragged_lengths(99, a.batch, 2, 8) if a.polyvore else a.items
    emb, mask = synth.outfit_batch(99, a.batch, a.pad, n_items)
    emb, mask = torch.from_numpy(emb).cuda(), torch.from_numpy(mask).cuda()
    labels = (torch.from_numpy(np.random.default_rng(99).random(a.batch) < 0.5).float() if a.polyvore else (torch.arange(a.batch) % 2).float()).cuda()
    loss_fn = FocalLoss(alpha=0.75, gamma=2, reduction="mean")

    def fwd_bwd():
        y = m(task=CP, outfit_embedding=emb, outfit_mask=mask).squeeze(-1)
        loss = loss_fn(y_hat=y, y_true=labels)
        loss.backward()
        return loss

    def step():
        opt.zero_grad(set_to_none=True)
        fwd_bwd()
        torch.nn.utils.clip_grad_norm_(params, max_norm=1.0)
        opt.step()

    def fwd_only():
        eng = m._engine(a.precision)
        eng.cp_train_fwd(emb, mask)

    what = "2..8 items (uniform)" if a.polyvore else f"{a.items} items"
    res = {"workload": f"CP trainer step, {a.batch} outfits x {what} (padded {a.pad}), precomputed embeddings",
           "precision": a.precision}
    res["ms_step"] = timed(step, a.steps, a.warmup)
    res["ms_fwd_bwd"] = timed(lambda: (opt.zero_grad(set_to_none=True), fwd_bwd()), a.steps, a.warmup)
    res["ms_tape_fwd"] = timed(fwd_only, a.steps, a.warmup)
    res["outfits_per_s"] = a.batch / res["ms_step"] * 1e3
    # the DP trainer's step (world size from the launcher; 1 = no collective)
    import torch.distributed as dist
    from outfitx_amd.trainer import CPTrainConfig, CPTrainer
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl")
    opt.zero_grad(set_to_none=True)
    batch = {"input_dict": {"task": CP, "outfit_embedding": emb, "outfit_mask": mask}, "label": labels}
    for acc in (1, 4):
        tr = CPTrainer(m, steps_per_epoch=10 ** 9, cfg=CPTrainConfig(accumulation_steps=acc), params=params)
        k = [0]
        def dp_step():
            tr.micro_step(batch, k[0]); k[0] += 1
        res["ms_step_dp" if acc == 1 else "ms_step_dp_accum4"] = timed(dp_step, a.steps if acc == 1 else 4 * max(a.steps // 4, 1), a.warmup if acc == 1 else 4)
    res["world"] = world
    res["outfits_per_s_dp"] = world * a.batch / res["ms_step_dp"] * 1e3
    rows = int(np.sum(n_items) + a.batch) if a.polyvore else a.batch * (a.items + 1)
    D, Fp = 1024, 2048
    res["gemm_tflop_per_step"] = 3 * 2 * rows * (4 * D * D + 2 * Fp * D) * 6 / 1e12
    res["tflops"] = res["gemm_tflop_per_step"] / (res["ms_fwd_bwd"] * 1e-3)

    if a.eager:
        layer = torch.nn.TransformerEncoderLayer(d_model=1024, nhead=16, dim_feedforward=2024, dropout=0.0, batch_first=True,
                                                 norm_first=True, activation=torch.nn.functional.mish)
        enc = torch.nn.TransformerEncoder(layer, num_layers=6, enable_nested_tensor=False).cuda().train()
        tok = torch.nn.Parameter(torch.randn(1024, device="cuda") * 0.02)
        head = torch.nn.Linear(1024, 1).cuda()
        eparams = list(enc.parameters()) + [tok] + list(head.parameters())
        eopt = torch.optim.AdamW(eparams, lr=2e-5)
        km = torch.cat([torch.zeros(a.batch, 1, dtype=torch.bool, device="cuda"), mask], 1)

        def eager_step():
            eopt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                x = torch.cat([tok.view(1, 1, -1).expand(a.batch, 1, -1), emb], 1)
                y = head(enc(x, src_key_padding_mask=km)[:, 0]).squeeze(-1)
                ce = torch.nn.functional.binary_cross_entropy_with_logits(y.float(), labels, reduction="none")
                p = torch.sigmoid(y.float())
                pt = p * labels + (1 - p) * (1 - labels)
                loss = ((0.75 * labels + 0.25 * (1 - labels)) * ce * (1 - pt) ** 2).mean()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(eparams, max_norm=1.0)
            eopt.step()

        res["ms_step_torch_eager_bf16"] = timed(eager_step, a.steps, a.warmup)
        res["speedup_vs_torch_eager"] = res["ms_step_torch_eager_bf16"] / res["ms_step"]
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
