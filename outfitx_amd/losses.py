"""FocalLoss — drop-in for the reference's `src.losses.FocalLoss` (src/losses/focal_loss.py:8-41), fused: loss value and
d loss / d logits come from ONE kernel (ofx_focal_loss_ex; reduction 'mean' | 'sum' | 'none' as focal_loss.py:36-41) instead of
~12 eager elementwise launches.  Used by the CP
trainer as `FocalLoss(alpha=0.75, gamma=2, reduction='mean')` (compatibility_prediction_trainer.py:369-370).
No CPU path: HIP tensors only."""
from __future__ import annotations

import torch
from torch import nn

from .engine import focal_loss


class _FocalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_hat, y_true, alpha, gamma, reduction):
        loss, dl = focal_loss(y_hat, y_true, alpha, gamma, 1.0, need_grad=y_hat.requires_grad, reduction=reduction)
        ctx.save_for_backward(dl)
        ctx.shape = y_hat.shape
        return loss.view(y_hat.shape) if reduction == "none" else loss

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return (dl * g.reshape(-1) if g.dim() else dl * g).view(ctx.shape), None, None, None, None


class FocalLoss(nn.Module):
    def __init__(self, gamma=2, alpha=0.5, reduction="mean"):
        super().__init__()
        assert gamma >= 0, f"Invalid Value for arg 'gamma': '{gamma}' \n Gamma should be non-negative"
        assert 0 <= alpha <= 1, f"Invalid Value for arg 'alpha': '{alpha}' \n Alpha should be in range [0, 1]"
        assert reduction in ["none", "mean", "sum"], f"Invalid Value for arg 'reduction': '{reduction}'"
        self.gamma, self.alpha, self.reduction = gamma, alpha, reduction

    def forward(self, y_hat: torch.Tensor, y_true: torch.Tensor) -> torch.Tensor:
        return _FocalFn.apply(y_hat, y_true, float(self.alpha), float(self.gamma), self.reduction)


class SetWiseRankingLoss(nn.Module):
    """Drop-in for the reference's `src.losses.SetWiseRankingLoss` (src/losses/set_wise_ranking_loss.py:5-39), the CIR trainer's
    loss (complementary_item_retrieval_trainer.py:79-88): with d+ = ||y_hat - y||, d-_k = ||y_hat - neg_k||,
        L_all  = sum over valid negatives of relu(d+ - d-_k + margin) / max(#valid, 1)
        L_hard = mean_b relu(d+ - min_k d-_k + margin)            (padded negatives count as +inf)
    returned as L_all + L_hard.  Caller-side torch code on [B,D] / [B,K,D] tensors (autograd supplies d loss / d y_hat, which
    the HIP backward of the CIR path consumes); `pairwise_distance`'s eps = 1e-6 is kept."""

    def __init__(self, margin: float = 2.0):
        super().__init__()
        self.margin = margin

    def forward(self, batch_y, batch_y_hat, batch_negative_samples, batch_negative_mask):
        d_pos = torch.linalg.vector_norm(batch_y_hat - batch_y + 1e-6, dim=-1)                 # F.pairwise_distance adds eps to the difference
        d_neg = torch.linalg.vector_norm(batch_y_hat[:, None, :] - batch_negative_samples, dim=-1)
        valid = ~batch_negative_mask
        n_valid = valid.sum().clamp(min=1)
        hinge_all = torch.relu(d_pos[:, None] - d_neg + self.margin) * valid
        hardest = d_neg.masked_fill(batch_negative_mask, float("inf")).amin(dim=1)
        return hinge_all.sum() / n_valid + torch.relu(d_pos - hardest + self.margin).mean()
