"""CP training step on a real MI355X: forward with a tape + hand-written backward, through
`src.models.OutfitX` in train() mode (autograd.Function over ofx_cp_train_fwd / ofx_cp_train_bwd).

Pinned by tests/golden/train_step.npz: the REFERENCE model in train() mode (dropout 0, fp32) on a seeded ragged batch —
loss, logits, per-parameter gradient norms, full small gradients, strided samples of the big ones, the clip norm and
post-AdamW parameters (oracle/gen_golden.py §8).

Tolerances: the step computes with single-product MFMA operands like the reference's autocast training.
  f16  : per-parameter ||g - g_ref|| / ||g_ref|| <= 5e-3, loss 1e-3
  bf16 : per-parameter                           <= 3e-2, loss 1e-2
"""
import warnings

import numpy as np
import pytest
import torch

from conftest import W_SEED, golden
from outfitx_amd import synth

pytestmark = pytest.mark.gpu
warnings.simplefilter("ignore")


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make_model(train_precision, dropout=0.0):
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from src.models import OutfitX
    from src.models.configs import ItemEncoderConfig, OutfitXConfig
    cfg = OutfitXConfig(item_encoder=ItemEncoderConfig(type="clip"))
    cfg.transformer.dropout = dropout
    m = OutfitX(cfg, train_precision=train_precision)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.full_state_dict(W_SEED).items()}, strict=True)
    return m.cuda().train()


def trainable(m):
    return {k: v for k, v in m.named_parameters() if not k.startswith("item_encoder.")}


def nrm(a):
    return float(np.sqrt((np.asarray(a, np.float64) ** 2).sum()))


@pytest.mark.parametrize("prec,gtol,ltol", [("f16", 5e-3, 1e-3), ("bf16", 3e-2, 1e-2)])
@pytest.mark.parametrize("fused_loss", [True, False])
def test_cp_train_step_vs_reference_golden(prec, gtol, ltol, fused_loss):
    from src.losses import FocalLoss
    from src.models.datatypes import OutfitCompatibilityPredictionTask as CP
    g = golden("train_step")
    n = g["n_items"]
    emb, mask = synth.outfit_batch(int(g["seed"]), len(n), 16, n)
    assert synth.checksum(emb) == str(g["emb_crc"])
    m = make_model(prec)
    params = trainable(m)
    opt = torch.optim.AdamW(list(params.values()), lr=float(g["lr"]))
    opt.zero_grad()
    y_hat = m(task=CP, outfit_embedding=cu(emb), outfit_mask=cu(mask)).squeeze(-1)
    labels = cu(g["labels"])
    if fused_loss:
        loss = FocalLoss(alpha=0.75, gamma=2, reduction="mean")(y_hat=y_hat, y_true=labels)
    else:       # the reference's composition of torch ops (focal_loss.py:26-41) on our logits
        ce = torch.nn.functional.binary_cross_entropy_with_logits(y_hat, labels, reduction="none")
        p = torch.sigmoid(y_hat)
        pt = p * labels + (1 - p) * (1 - labels)
        loss = ((0.75 * labels + 0.25 * (1 - labels)) * ce * (1 - pt) ** 2).mean()
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) <= ltol * abs(float(g["loss"]))
    assert np.abs(y_hat.detach().cpu().numpy() - g["logits"]).max() <= ltol * max(1.0, np.abs(g["logits"]).max())
    # parameters the reference gives no gradient to get none here either
    for k in g["no_grad_names"]:
        assert params[str(k)].grad is None, k
    worst = {}
    for k, ref_norm in zip(g["grad_names"], g["grad_norms"]):
        k = str(k)
        gr = params[k].grad
        assert gr is not None and gr.shape == params[k].shape, k
        gr = gr.detach().cpu().numpy()
        if "grad/" + k in g.files:
            err = nrm(gr - g["grad/" + k]) / max(ref_norm, 1e-30)
        else:
            ref = g["gsample/" + k]
            err = nrm(gr.ravel()[::1009] - ref) / max(nrm(ref), 1e-30)
            assert abs(nrm(gr) - ref_norm) <= gtol * ref_norm, (k, nrm(gr), ref_norm)
        worst[k] = err
    bad = {k: v for k, v in worst.items() if not v <= gtol}
    assert not bad, bad
    clip = float(torch.nn.utils.clip_grad_norm_(list(params.values()), max_norm=1.0))
    assert abs(clip - float(g["clip_norm"])) <= gtol * float(g["clip_norm"])
    opt.step()
    for k in ("outfit_token", "cp_ffn.1.weight", "transformer_encoder.layers.0.norm1.weight"):
        got = params[k].detach().cpu().numpy()
        before = synth.full_state_dict(W_SEED)[k]
        # AdamW's first step moves every weight by ~lr * sign(g): compare the UPDATE, not the weight.  An element whose
        # gradient is ~0 may flip sign under operand rounding, so bound the FRACTION of disagreeing elements.
        du, dr = got - before, g["post/" + k] - before
        assert np.mean(np.abs(du - dr) > 0.1 * float(g["lr"])) <= 0.01, k


def test_train_step_matches_torch_autograd_of_the_same_module():
    """Independent check on fresh inputs: the same nn.TransformerEncoder (plain PyTorch fp32, run on the GPU box's CPU)
    with the same weights -> autograd gradients; ours (f16 operands) within 5e-3 per parameter."""
    from src.models.datatypes import OutfitCompatibilityPredictionTask as CP
    n = np.array([2, 9, 16, 1, 5, 7])
    emb, mask = synth.outfit_batch(4321, len(n), 16, n)
    m = make_model("f16")
    up = torch.linspace(-1.0, 2.0, len(n))
    y = m(task=CP, outfit_embedding=cu(emb), outfit_mask=cu(mask))
    (y.squeeze(-1) * up.cuda()).sum().backward()
    ours = {k: v.grad.detach().cpu() for k, v in trainable(m).items() if v.grad is not None}
    # plain torch reference of the same op
    t = m.cfg.transformer
    layer = torch.nn.TransformerEncoderLayer(d_model=1024, nhead=t.n_head, dim_feedforward=t.d_ffn, dropout=0.0, batch_first=True,
                                             norm_first=True, activation=torch.nn.functional.mish)
    enc = torch.nn.TransformerEncoder(layer, num_layers=t.n_layers, enable_nested_tensor=False)
    enc.load_state_dict({k: v.detach().cpu() for k, v in m.transformer_encoder.state_dict().items()})
    tok = m.outfit_token.detach().cpu().clone().requires_grad_(True)
    w = m.cp_ffn[1].weight.detach().cpu().clone().requires_grad_(True)
    b = m.cp_ffn[1].bias.detach().cpu().clone().requires_grad_(True)
    x = torch.cat([tok.view(1, 1, -1).expand(len(n), 1, -1), torch.from_numpy(emb)], 1)
    km = torch.cat([torch.zeros(len(n), 1, dtype=torch.bool), torch.from_numpy(mask)], 1)
    out = enc.train()(x, src_key_padding_mask=km)[:, 0]
    ((out @ w.t() + b).squeeze(-1) * up).sum().backward()
    ref = {"outfit_token": tok.grad, "cp_ffn.1.weight": w.grad, "cp_ffn.1.bias": b.grad}
    ref.update({"transformer_encoder." + k: v.grad for k, v in enc.named_parameters()})
    assert set(ref) == set(ours)
    bad = {}
    for k in ref:
        e = nrm((ours[k] - ref[k]).numpy()) / max(nrm(ref[k].numpy()), 1e-30)
        if not e <= 5e-3:
            bad[k] = e
    assert not bad, bad


def test_fused_focal_loss_and_gradient_vs_golden_and_torch():
    from src.losses import FocalLoss
    g = golden("aux")
    x = cu(g["focal_logits"]).requires_grad_(True)
    y = cu(g["focal_labels"])
    loss = FocalLoss(alpha=0.75, gamma=2.0)(x, y)
    assert abs(float(loss) - float(g["focal_value"])) <= 2e-6 * abs(float(g["focal_value"])) + 1e-7
    (loss * 3.0).backward()
    xr = torch.from_numpy(g["focal_logits"]).double().requires_grad_(True)
    yr = torch.from_numpy(g["focal_labels"]).double()
    ce = torch.nn.functional.binary_cross_entropy_with_logits(xr, yr, reduction="none")
    p = torch.sigmoid(xr)
    pt = p * yr + (1 - p) * (1 - yr)
    (((0.75 * yr + 0.25 * (1 - yr)) * ce * (1 - pt) ** 2).mean() * 3.0).backward()
    assert np.abs(x.grad.cpu().numpy() - xr.grad.numpy()).max() <= 1e-6 * np.abs(xr.grad.numpy()).max() + 1e-9


def test_training_guards():
    from src.models.datatypes import (OutfitCompatibilityPredictionTask as CP, OutfitComplementaryItemRetrievalTask as CIR)
    emb, mask = synth.outfit_batch(5, 2, 4, 3)
    m = make_model("bf16", dropout=0.3)
    with pytest.raises(NotImplementedError):
        m(task=CP, outfit_embedding=cu(emb), outfit_mask=cu(mask))
    with torch.no_grad():                      # scoring in train() mode under no_grad stays available
        m(task=CP, outfit_embedding=cu(emb), outfit_mask=cu(mask))
    m.cfg.transformer.dropout = 0.0
    with pytest.raises(NotImplementedError):
        m(task=CIR, outfit_embedding=cu(emb), outfit_mask=cu(mask), target_item_text_embedding=cu(synth.unit_rows(5, "t", 2, 512)))
    m.train_precision = "bf16x3"
    with pytest.raises(ValueError):
        m(task=CP, outfit_embedding=cu(emb), outfit_mask=cu(mask))
