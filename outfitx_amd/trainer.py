"""Data-parallel CP training loop over the HIP training step (BASELINE config 5).

Counterpart of the reference's `CompatibilityPredictionTrainer.train_epoch` step
(src/trains/trainers/compatibility_prediction_trainer.py:57-81) and its `build_metrics` all-gather (:372-405), written
for one process per GPU over RCCL/xGMI rather than DistributedDataParallel:

  forward (tape, bf16 operands) -> FocalLoss(.75, 2, mean) -> / accumulation_steps -> backward (hand-written HIP)
  every `accumulation_steps` micro-batches:  ONE all-reduce of the flat gradient arena (mean over ranks)
                                             -> clip_grad_norm_(1.0) -> AdamW(lr) -> OneCycleLR step -> zero
  epoch end: all-gather logits / labels / loss -> AUC, accuracy, precision, recall, F1 (same formulas as :406-436).

Differences from DDP by design (MI355X / xGMI, point-to-point links, per-link bound rings):
  * gradients live in ONE contiguous fp32 arena (`FlatGrads`; every p.grad is a view): the clip norm is one reduction over the
    arena instead of 75 per-tensor norms, and the all-reduce goes out as one contiguous slice per encoder layer (34 MB each,
    6 + 1 collectives for the 205 MB) STARTED WHILE THE BACKWARD STILL RUNS: `ofx_train_arm_layer_events` makes the hand-written
    backward record a HIP event when layer l's gradients are final (layers finish last-to-first), and the slice's RCCL
    all-reduce is enqueued on a side stream behind that event - 5/6 of the reduction overlaps the backward of the layers below
    (xGMI is point-to-point: 34 MB slices are still bandwidth-, not latency-bound per link);
  * the all-reduce runs once per OPTIMIZER step; DDP without no_sync() (what the reference does) reduces on every
    micro-batch, 4x the traffic at accumulation_steps = 4;
  * no per-step all_gather_object / barrier pair (reference C4/C5): error propagation is the job of the
    launcher (torchrun tears the group down when a rank raises).

The loop itself is device-agnostic torch host code (the CPU tests drive it with a stub module over gloo); the model it is
meant for, `outfitx_amd.OutfitX` in train() mode, only runs on a HIP device.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


@dataclass
class CPTrainConfig:
    """Defaults = the reference's CompatibilityPredictionTrainConfig / BaseTrainConfig (src/trains/configs)."""
    learning_rate: float = 2e-5
    accumulation_steps: int = 4
    n_epochs: int = 200
    max_grad_norm: float = 1.0
    focal_alpha: float = 0.75
    focal_gamma: float = 2.0
    pct_start: float = 0.3
    div_factor: float = 25.0
    final_div_factor: float = 1e4
    fused_optimizer: bool = True      # torch.optim.AdamW(fused=True): one multi-tensor kernel per step


class FlatGrads:
    """One contiguous fp32 arena holding every trainable parameter's gradient; p.grad are views into it, so autograd
    accumulates in place and reductions / norms / zeroing are single kernels over the arena."""

    def __init__(self, params: Sequence[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        dev = self.params[0].device
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 63) // 64 * 64
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            p.grad = self.flat[o:o + p.numel()].view_as(p)

    def zero_(self):
        self.flat.zero_()
        for p, o in zip(self.params, self.offsets):      # re-attach views an optimizer's zero_grad(set_to_none=True) dropped
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * o:
                p.grad = self.flat[o:o + p.numel()].view_as(p)

    def all_reduce_mean_(self, group=None):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(dist.get_world_size(group))

    def slices(self, groups: Sequence[Sequence[torch.nn.Parameter]]):
        """groups: lists of parameters, each list contiguous in the arena (an encoder layer's 12 tensors) -> ([(lo, hi)] per group,
        [(lo, hi)] of everything not covered), in floats."""
        off = {id(p): (o, o + (p.numel() + 63) // 64 * 64) for p, o in zip(self.params, self.offsets)}
        out = []
        for g in groups:
            rs = sorted(off[id(p)] for p in g if id(p) in off)
            if not rs or any(a[1] != b[0] for a, b in zip(rs, rs[1:])):
                raise ValueError("a gradient bucket must be a contiguous run of the arena")
            out.append((rs[0][0], rs[-1][1]))
        rest, cur = [], 0
        for lo, hi in sorted(out):
            if lo > cur:
                rest.append((cur, lo))
            cur = max(cur, hi)
        if cur < self.flat.numel():
            rest.append((cur, self.flat.numel()))
        return out, rest

    def clip_norm_(self, max_norm: float) -> torch.Tensor:
        """torch.nn.utils.clip_grad_norm_ semantics (L2, eps 1e-6, coefficient clamped to 1); stays on the device."""
        total = torch.linalg.vector_norm(self.flat)
        self.flat.mul_(torch.clamp(max_norm / (total + 1e-6), max=1.0))
        return total


def cp_metrics(y_hats: torch.Tensor, labels: torch.Tensor) -> Dict[str, float]:
    """Accuracy / Precision / Recall / F1 / AUC of compatibility logits (cp_trainer:406-436).  AUC by the rank statistic
    (ties get the average rank) — what sklearn.metrics.roc_auc_score returns — without leaving torch."""
    probs = torch.sigmoid(y_hats.float()).detach().cpu().double()
    lab = labels.detach().cpu().int()
    n_pos, n_neg = int((lab == 1).sum()), int((lab == 0).sum())
    if n_pos and n_neg:
        order = torch.argsort(probs)
        sp = probs[order]
        ranks = torch.arange(1, len(sp) + 1, dtype=torch.float64)
        uniq, inv, cnt = torch.unique_consecutive(sp, return_inverse=True, return_counts=True)
        ends = torch.cumsum(cnt, 0).double()
        avg = ends - (cnt.double() - 1) / 2
        ranks = avg[inv]
        r = torch.empty_like(ranks); r[order] = ranks
        auc = float((r[lab == 1].sum() - n_pos * (n_pos + 1) / 2) / (n_pos * n_neg))
    else:
        auc = 0.0
    pred = (probs > 0.5).int()
    tp = int(((pred == 1) & (lab == 1)).sum()); fp = int(((pred == 1) & (lab == 0)).sum()); fn = int(((pred == 0) & (lab == 1)).sum())
    precision = tp / (tp + fp) if tp + fp else 0.0
    recall = tp / (tp + fn) if tp + fn else 0.0
    f1 = 2 * precision * recall / (precision + recall) if precision + recall else 0.0
    return {"Accuracy": float((pred == lab).float().mean()), "Precision": precision, "Recall": recall, "F1": f1, "AUC": auc}


def gather_epoch(local_y: torch.Tensor, local_labels: torch.Tensor, local_loss: torch.Tensor, batch_count: int, group=None):
    """cp_trainer:372-405: all-gather every rank's epoch logits / labels / summed loss -> global metrics on every rank.
    Ranks may hold different sample counts (the reference assumes equal ones): sizes are gathered first."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if world > 1:
        n = torch.tensor([local_y.numel()], dtype=torch.int64, device=local_y.device)
        sizes = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(sizes, n, group=group)
        mx = int(max(int(s) for s in sizes))
        def gat(t):
            pad = torch.zeros(mx, dtype=t.dtype, device=t.device); pad[: t.numel()] = t.reshape(-1)
            out = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(out, pad, group=group)
            return torch.cat([o[: int(s)] for o, s in zip(out, sizes)])
        ys, ls = gat(local_y.detach().float()), gat(local_labels.detach().float())
        losses = [torch.empty_like(local_loss) for _ in range(world)]
        dist.all_gather(losses, local_loss.detach(), group=group)
        loss = torch.stack(losses).mean() / batch_count
    else:
        ys, ls, loss = local_y.detach().float(), local_labels.detach().float(), local_loss.detach() / batch_count
    return {"loss": float(loss), **cp_metrics(ys, ls)}


class CPTrainer:
    """model(task=CP, outfit_embedding=..., outfit_mask=...) -> [B,1] logits; batches are dicts like the reference's
    collate output: {'input_dict': {'task', 'outfit_embedding', 'outfit_mask'}, 'label'}."""

    def __init__(self, model: torch.nn.Module, steps_per_epoch: int, cfg: Optional[CPTrainConfig] = None,
                 loss_fn: Optional[Callable] = None, params: Optional[Iterable[torch.nn.Parameter]] = None, group=None):
        self.model, self.cfg, self.group = model, cfg or CPTrainConfig(), group
        c = self.cfg
        ps = list(params) if params is not None else [p for p in model.parameters() if p.requires_grad]
        self.grads = FlatGrads(ps)
        if loss_fn is None:
            from .losses import FocalLoss
            loss_fn = FocalLoss(alpha=c.focal_alpha, gamma=c.focal_gamma, reduction="mean")
        self.loss_fn = loss_fn
        dev = self.grads.flat.device
        fused = c.fused_optimizer and dev.type == "cuda"
        self.optimizer = torch.optim.AdamW(self.grads.params, lr=c.learning_rate, **({"fused": True} if fused else {}))
        self.scheduler = torch.optim.lr_scheduler.OneCycleLR(
            optimizer=self.optimizer, max_lr=c.learning_rate, epochs=c.n_epochs,
            steps_per_epoch=math.ceil(steps_per_epoch / c.accumulation_steps), pct_start=c.pct_start, anneal_strategy="cos",
            div_factor=c.div_factor, final_div_factor=c.final_div_factor)
        self.steps_per_epoch = steps_per_epoch
        self.last_grad_norm: Optional[torch.Tensor] = None
        # outfitx_amd.OutfitX: let the backward kernels add straight into the arena views (no per-tensor accumulate kernels);
        # valid because nothing here relies on autograd hooks (DDP would)
        if hasattr(model, "_cp_train_forward"):
            model.grad_sink = True
        # per-layer gradient slices for the overlapped reduction (None: the model has no encoder-layer structure -> one all-reduce)
        self.layer_slices = self.rest_slices = None
        layers = getattr(getattr(model, "transformer_encoder", None), "layers", None)
        if layers is not None:
            try:
                self.layer_slices, self.rest_slices = self.grads.slices([[p for p in l.parameters() if p.requires_grad] for l in layers])
            except ValueError:
                self.layer_slices = None
        self.overlap_reduce = True
        self._layer_events = None
        self._comm_stream = None

    def _world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def _arm_overlap(self) -> bool:
        """Before the backward of an accumulation window's LAST micro-batch: arm the per-layer completion events (HIP model only)."""
        if not (self.overlap_reduce and self.layer_slices and self._world() > 1 and hasattr(self.model, "arm_bwd_layer_events")
                and self.grads.flat.is_cuda):
            return False
        # the events mean "final in the arena" only on the gradient-sink path; any tensor whose .grad is not the arena view (detached,
        # non-contiguous, not fp32) sends the backward through autograd's accumulation, which runs AFTER the events: plain reduction then
        if hasattr(self.model, "sink_ready") and not self.model.sink_ready():
            return False
        if self._layer_events is None:
            self._layer_events = [torch.cuda.Event() for _ in self.layer_slices]
            for e in self._layer_events:
                e.record()                                   # materialises the HIP handle
            self._comm_stream = torch.cuda.Stream(self.grads.flat.device)
        self.model.arm_bwd_layer_events(self._layer_events)
        return True

    def _reduce_mean(self, overlapped: bool) -> None:
        """Mean of the gradient arena over the ranks.  overlapped: layer l's slice is reduced on the side stream as soon as its event
        fires (the backward of the layers below is still running on the main stream); else slice by slice after the backward (CPU /
        gloo, stub models) or as one collective when the model has no layer structure.  Same sums either way."""
        world = self._world()
        if world <= 1:
            return
        flat = self.grads.flat
        if not self.layer_slices:
            self.grads.all_reduce_mean_(self.group)
            return
        works = []
        if overlapped:
            main, comm = torch.cuda.current_stream(flat.device), self._comm_stream
            for l in reversed(range(len(self.layer_slices))):
                lo, hi = self.layer_slices[l]
                comm.wait_event(self._layer_events[l])
                with torch.cuda.stream(comm):
                    works.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            comm.wait_stream(main)                           # heads / tokens: final only when the whole backward is done
            with torch.cuda.stream(comm):
                for lo, hi in self.rest_slices:
                    works.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            for w in works:
                w.wait()
            main.wait_stream(comm)
        else:
            for lo, hi in list(reversed(self.layer_slices)) + list(self.rest_slices):
                dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
        flat.div_(world)

    def micro_step(self, batch: dict, step: int):
        """One micro-batch: forward, loss / accumulation_steps, backward; optimizer step on the accumulation boundary.
        Returns (detached loss, detached logits)."""
        c = self.cfg
        dev = self.grads.flat.device
        inp = {k: (v if k == "task" else v.to(dev, non_blocking=True)) for k, v in batch["input_dict"].items()}
        labels = batch["label"].to(dev, non_blocking=True)
        y_hat = self.model(**inp).squeeze(dim=-1)
        loss = self.loss_fn(y_hat=y_hat, y_true=labels)
        boundary = (step + 1) % c.accumulation_steps == 0 or step + 1 == self.steps_per_epoch
        overlapped = boundary and self._arm_overlap()
        (loss / c.accumulation_steps).backward()
        if boundary:
            self._reduce_mean(overlapped)
            self.last_grad_norm = self.grads.clip_norm_(c.max_grad_norm)
            self.optimizer.step()
            self.scheduler.step()
            self.grads.zero_()
            getattr(self.model, "mark_weights_changed", lambda: None)()     # fused optimizers do not bump tensor versions
        return loss.detach(), y_hat.detach(), labels

    def train_epoch(self, batches: Iterable[dict]) -> Dict[str, float]:
        self.model.train()
        self.grads.zero_()
        total = torch.zeros((), dtype=torch.float32, device=self.grads.flat.device)
        ys: List[torch.Tensor] = []; ls: List[torch.Tensor] = []
        n = 0
        for step, batch in enumerate(batches):
            loss, y, lab = self.micro_step(batch, step)
            total += loss; ys.append(y); ls.append(lab); n += 1
        return gather_epoch(torch.cat(ys), torch.cat(ls), total, max(n, 1), self.group)
