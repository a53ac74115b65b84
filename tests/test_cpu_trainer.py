"""CPU checks of the data-parallel CP training loop (outfitx_amd/trainer.py): the loop's host logic —
flat gradient arena, one all-reduce per optimizer step, clip, AdamW + OneCycleLR, epoch metric all-gather — driven by a
small plain-torch stand-in module over a world_size-2 gloo group and compared with a single-process run of the same
global batches.  (The real model only runs on a HIP device; its step is covered by tests/test_gpu_train.py.)"""
import os
import socket
import subprocess
import sys

import numpy as np
import torch

from conftest import ROOT


def test_cp_metrics_match_sklearn():
    from sklearn.metrics import roc_auc_score
    from outfitx_amd.trainer import cp_metrics
    g = np.random.default_rng(0)
    y = torch.from_numpy(np.round(g.standard_normal(500), 1).astype(np.float32))      # rounded -> ties
    lab = torch.from_numpy((g.random(500) < 0.4).astype(np.float32))
    m = cp_metrics(y, lab)
    p = torch.sigmoid(y).numpy()
    assert abs(m["AUC"] - roc_auc_score(lab.numpy().astype(int), p)) < 1e-12
    pred = p > 0.5
    tp, fp, fn = (pred & (lab.numpy() == 1)).sum(), (pred & (lab.numpy() == 0)).sum(), (~pred & (lab.numpy() == 1)).sum()
    assert abs(m["Precision"] - tp / (tp + fp)) < 1e-12 and abs(m["Recall"] - tp / (tp + fn)) < 1e-12
    assert cp_metrics(torch.zeros(4), torch.ones(4))["AUC"] == 0.0          # single class -> 0.0 like cp_trainer:416


def test_flat_grads_clip_matches_torch_clip_grad_norm():
    from outfitx_amd.trainer import FlatGrads
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 1))
    ref = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 1))
    ref.load_state_dict(lin.state_dict())
    fg = FlatGrads(list(lin.parameters()))
    x = torch.randn(9, 7)
    lin(x).pow(2).sum().backward(); ref(x).pow(2).sum().backward()
    lin(x).sum().backward(); ref(x).sum().backward()                      # accumulation lands in the arena views
    want = torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
    got = fg.clip_norm_(1.0)
    assert torch.allclose(got, want, rtol=1e-6)
    for a, b in zip(lin.parameters(), ref.parameters()):
        assert a.grad.data_ptr() >= fg.flat.data_ptr() and torch.allclose(a.grad, b.grad, rtol=1e-6, atol=1e-8)
    torch.optim.AdamW(lin.parameters()).zero_grad(set_to_none=True)
    fg.zero_()
    assert all(p.grad is not None and not p.grad.any() for p in lin.parameters())


_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["OFX_ROOT"])
from outfitx_amd.trainer import CPTrainer, CPTrainConfig

class Stub(torch.nn.Module):                 # stands in for OutfitX: same call signature, plain torch, CPU
    def __init__(self):
        super().__init__()
        torch.manual_seed(1)
        self.a = torch.nn.Linear(16, 8); self.b = torch.nn.Linear(8, 1)
        # an encoder-layer structure like OutfitX's: the trainer reduces the gradients slice by slice, one per layer + the rest
        self.transformer_encoder = torch.nn.Module()
        self.transformer_encoder.layers = torch.nn.ModuleList([torch.nn.Linear(16, 16) for _ in range(3)])
    def forward(self, task, outfit_embedding, outfit_mask):
        keep = (~outfit_mask).float().unsqueeze(-1)
        x = outfit_embedding
        for l in self.transformer_encoder.layers:
            x = x + torch.tanh(l(x))
        pooled = (x * keep).sum(1) / keep.sum(1).clamp(min=1)
        return self.b(torch.nn.functional.mish(self.a(pooled)))

def focal(y_hat, y_true):                    # src/losses/focal_loss.py:26-41 in torch ops (the fused kernel needs a HIP device)
    ce = torch.nn.functional.binary_cross_entropy_with_logits(y_hat, y_true, reduction="none")
    p = torch.sigmoid(y_hat); pt = p * y_true + (1 - p) * (1 - y_true)
    return ((0.75 * y_true + 0.25 * (1 - y_true)) * ce * (1 - pt) ** 2).mean()

def batches(lo, hi, n_steps, bsz):
    g = np.random.default_rng(5)
    out = []
    for s in range(n_steps):
        emb = torch.from_numpy(g.standard_normal((bsz, 6, 16)).astype(np.float32))
        n = g.integers(1, 7, bsz)
        mask = torch.from_numpy(np.arange(6)[None, :] >= n[:, None])
        lab = torch.from_numpy((g.random(bsz) < 0.5).astype(np.float32))
        out.append({"input_dict": {"task": None, "outfit_embedding": emb[lo:hi], "outfit_mask": mask[lo:hi]}, "label": lab[lo:hi]})
    return out

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
cfg = CPTrainConfig(learning_rate=1e-2, accumulation_steps=2, n_epochs=2)
STEPS, BSZ = 5, 8                            # 5 micro-steps: the last optimizer step closes a short accumulation window
half = BSZ // world
m = Stub()
tr = CPTrainer(m, steps_per_epoch=STEPS, cfg=cfg, loss_fn=focal)
assert tr.layer_slices is not None and len(tr.layer_slices) == 3 and tr.rest_slices       # per-layer slices + the head parameters
covered = sorted(tr.layer_slices + tr.rest_slices)
assert covered[0][0] == 0 and covered[-1][1] == tr.grads.flat.numel() and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
metrics = None
for ep in range(2):
    metrics = tr.train_epoch(batches(rank * half, (rank + 1) * half, STEPS, BSZ))
# single-process reference on the full global batches (no process group use: group of size 1 semantics)
ref = Stub()
opt = torch.optim.AdamW(ref.parameters(), lr=1e-2)
sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-2, epochs=2, steps_per_epoch=3, pct_start=0.3, anneal_strategy="cos", div_factor=25, final_div_factor=1e4)
ys, ls, tot = [], [], 0.0
for ep in range(2):
    opt.zero_grad(); ys, ls, tot = [], [], 0.0
    for step, b in enumerate(batches(0, BSZ, STEPS, BSZ)):
        y = ref(**b["input_dict"]).squeeze(-1)
        loss = focal(y, b["label"]); (loss / 2).backward()
        tot += float(loss); ys.append(y.detach()); ls.append(b["label"])
        if (step + 1) % 2 == 0 or step + 1 == STEPS:
            torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0); opt.step(); sch.step(); opt.zero_grad()
for a, b in zip(m.parameters(), ref.parameters()):
    assert torch.allclose(a, b, rtol=2e-5, atol=1e-6), (a - b).abs().max()
assert abs(tr.scheduler.get_last_lr()[0] - sch.get_last_lr()[0]) < 1e-12
# epoch metrics: logits/labels of all ranks gathered -> same values on every rank as the single-process epoch
from outfitx_amd.trainer import cp_metrics
# the DP epoch metrics are computed from logits produced DURING the epoch; so are the reference's
want = cp_metrics(torch.cat(ys), torch.cat(ls))
# rank-sharded samples arrive in a different order, which the metrics do not depend on
for k in ("Accuracy", "Precision", "Recall", "F1", "AUC"):
    assert abs(metrics[k] - want[k]) < 1e-6, (k, metrics[k], want[k])
assert abs(metrics["loss"] - tot / STEPS) < 1e-5
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_dp_trainer_world2_gloo_matches_single_process(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OFX_ROOT=ROOT, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == 2
