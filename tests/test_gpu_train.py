"""CP training step on a real MI355X: forward with a tape + hand-written backward, through
`src.models.OutfitX` in train() mode (autograd.Function over ofx_cp_train_fwd / ofx_cp_train_bwd).

Pinned by tests/golden/train_step.npz: the REFERENCE model in train() mode (dropout 0, fp32) on a seeded ragged batch —
loss, logits, per-parameter gradient norms, full small gradients, strided samples of the big ones, the clip norm and
post-AdamW parameters (oracle/gen_golden.py §8).

Tolerances: the step computes with single-product MFMA operands like the reference's autocast training.
  f16  : per-parameter ||g - g_ref|| / ||g_ref|| <= 5e-3, loss / logits 2e-3
  bf16 : per-parameter                           <= 3e-2, loss / logits 1e-2
(q|k|v stay in the operand type between the in_proj GEMM and the attention, as under torch autocast.)
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


@pytest.mark.parametrize("prec,gtol,ltol", [("f16", 5e-3, 2e-3), ("bf16", 3e-2, 1e-2)])
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


@pytest.mark.parametrize("n,Lp", [([2, 9, 16, 1, 5, 7], 16), ([31, 1, 22, 30], 31)])
def test_train_step_matches_torch_autograd_of_the_same_module(n, Lp):
    """Independent check on fresh inputs: the same nn.TransformerEncoder (plain PyTorch fp32, run on the GPU box's CPU)
    with the same weights -> autograd gradients; ours (f16 operands) within 5e-3 per parameter.  The second case runs the
    longest sets the kernels take (31 items + prefix = 32 rows: the SMAX = 32 attention variants)."""
    from src.models.datatypes import OutfitCompatibilityPredictionTask as CP
    n = np.array(n)
    emb, mask = synth.outfit_batch(4321, len(n), Lp, n)
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


@pytest.mark.parametrize("reduction", ["sum", "none"])
def test_focal_loss_reductions_match_the_reference_composition(reduction):
    """FocalLoss(reduction='sum'|'none') (focal_loss.py:36-41) - value and gradient vs the reference's composition of torch ops in
    float64, on the golden logits."""
    from src.losses import FocalLoss
    g = golden("aux")
    x = cu(g["focal_logits"]).requires_grad_(True)
    y = cu(g["focal_labels"])
    loss = FocalLoss(alpha=0.75, gamma=2.0, reduction=reduction)(x, y)
    xr = torch.from_numpy(g["focal_logits"]).double().requires_grad_(True)
    yr = torch.from_numpy(g["focal_labels"]).double()
    ce = torch.nn.functional.binary_cross_entropy_with_logits(xr, yr, reduction="none")
    p = torch.sigmoid(xr)
    pt = p * yr + (1 - p) * (1 - yr)
    ref = (0.75 * yr + 0.25 * (1 - yr)) * ce * (1 - pt) ** 2
    wgt = torch.linspace(0.5, 2.0, xr.numel(), dtype=torch.float64).view(xr.shape)      # a non-uniform upstream gradient
    if reduction == "sum":
        ref = ref.sum()
        assert loss.dim() == 0
        (loss * 3.0).backward(); (ref * 3.0).backward()
    else:
        assert loss.shape == x.shape
        (loss * wgt.float().to(loss.device)).sum().backward(); (ref * wgt).sum().backward()
    assert np.abs(loss.detach().cpu().numpy() - ref.detach().numpy()).max() <= 3e-6 * np.abs(ref.detach().numpy()).max() + 1e-7
    assert np.abs(x.grad.cpu().numpy() - xr.grad.numpy()).max() <= 2e-6 * np.abs(xr.grad.numpy()).max() + 1e-9


def test_dropout_masks_are_stateless_and_have_the_right_rate():
    from outfitx_amd.engine import dropout_mask
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    a = dropout_mask(0.3, 1234, 5, 1000, 2048, torch.device("cuda"))
    b = dropout_mask(0.3, 1234, 5, 1000, 2048, torch.device("cuda"))
    c = dropout_mask(0.3, 1234, 6, 1000, 2048, torch.device("cuda"))
    d = dropout_mask(0.3, 1235, 5, 1000, 2048, torch.device("cuda"))
    assert torch.equal(a, b) and not torch.equal(a, c) and not torch.equal(a, d)
    vals = torch.unique(a).cpu().numpy()
    assert np.allclose(vals, [0.0, 1.0 / 0.7], rtol=1e-6)
    for t in (a, c, d):
        keep = float((t > 0).float().mean())
        assert abs(keep - 0.7) < 3e-3                                   # 2M samples: sigma = 3.2e-4
    # rows and columns are decorrelated: per-row and per-column keep rates scatter like a binomial
    rows = (a > 0).float().mean(1).cpu().numpy(); cols = (a > 0).float().mean(0).cpu().numpy()
    assert abs(rows.std() - np.sqrt(0.21 / 2048)) < 2e-3 and abs(cols.std() - np.sqrt(0.21 / 1000)) < 3e-3
    assert abs(float(((a > 0) & (c > 0)).float().mean()) - 0.49) < 3e-3    # sites are independent
    assert torch.equal(dropout_mask(0.0, 1, 0, 4, 8, torch.device("cuda")), torch.ones(4, 8, device="cuda"))


def test_dropout_training_step_matches_torch_autograd_with_the_same_masks():
    """Train-mode dropout 0.3 (the reference default, transformer_config.py:16) at all five kinds of site.  The masks are
    exported from the library and replayed in a float64 torch re-statement of the same network; logits and every
    parameter gradient must agree (f16 operands: 5e-3)."""
    from outfitx_amd.engine import dropout_mask
    from src.models.datatypes import OutfitCompatibilityPredictionTask as CP
    p = 0.3
    n = np.array([3, 8, 1, 5, 6])
    B, Lp = len(n), 8
    emb, mask = synth.outfit_batch(777, B, Lp, n)
    m = make_model("f16", dropout=p)
    torch.manual_seed(99)
    up = torch.linspace(-1.0, 2.0, B)
    y = m(task=CP, outfit_embedding=cu(emb), outfit_mask=cu(mask))
    (y.squeeze(-1) * up.cuda()).sum().backward()
    pp, seed = m.last_dropout
    assert pp == p
    torch.manual_seed(99)
    y2 = m(task=CP, outfit_embedding=cu(emb), outfit_mask=cu(mask))
    assert torch.equal(y, y2), "same torch seed -> same masks"
    ours = {k: v.grad.detach().cpu().double() for k, v in trainable(m).items() if v.grad is not None}
    dev = torch.device("cuda")
    S = n + 1
    cu_rows = np.concatenate([[0], np.cumsum(S)])
    M = int(cu_rows[-1])
    P = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in trainable(m).items()}
    nl, H, D, F = m.cfg.transformer.n_layers, 16, 1024, m.cfg.transformer.d_ffn
    mk = lambda site, r, c: dropout_mask(p, seed, site, r, c, dev).cpu().double()
    logits = []
    # rows of the non-attention masks: the global pad-free row, except in the LAST layer, whose out-proj / FFN only run on the
    # B prefix rows (compacted: mask row = outfit index; the other rows of that layer never reach the head)
    masks = {l: (mk(4 * l, B * H, 32 * 32).view(B, H, 32, 32), mk(4 * l + 1, M, D), mk(4 * l + 2, M, 2048)[:, :F], mk(4 * l + 3, M, D)) for l in range(nl - 1)}
    last = (mk(4 * (nl - 1), B * H, 32 * 32).view(B, H, 32, 32), mk(4 * (nl - 1) + 1, B, D), mk(4 * (nl - 1) + 2, B, 2048)[:, :F], mk(4 * (nl - 1) + 3, B, D))
    mh = mk(4 * nl, B, D)
    ln = torch.nn.functional.layer_norm
    for b in range(B):
        r0, s = int(cu_rows[b]), int(S[b])
        X = torch.cat([P["outfit_token"].view(1, D), torch.from_numpy(emb[b, :n[b]]).double()], 0)
        for l in range(nl):
            q = lambda name: P[f"transformer_encoder.layers.{l}.{name}"]
            if l < nl - 1:
                m0, m1, m2, m3 = masks[l]
                m1, m2, m3 = m1[r0:r0 + s], m2[r0:r0 + s], m3[r0:r0 + s]
            else:
                m0 = last[0]
                m1, m2, m3 = [torch.cat([t_[b:b + 1], torch.ones(s - 1, t_.shape[1], dtype=torch.float64)], 0) for t_ in last[1:]]
            h1 = ln(X, (D,), q("norm1.weight"), q("norm1.bias"), 1e-5)
            qkv = h1 @ q("self_attn.in_proj_weight").t() + q("self_attn.in_proj_bias")
            qh, kh, vh = [t.view(s, H, 64).transpose(0, 1) for t in qkv.split(D, dim=1)]
            pr = torch.softmax(qh @ kh.transpose(1, 2) / 8.0, -1) * m0[b, :, :s, :s]
            o = (pr @ vh).transpose(0, 1).reshape(s, D)
            X = X + (o @ q("self_attn.out_proj.weight").t() + q("self_attn.out_proj.bias")) * m1
            h2 = ln(X, (D,), q("norm2.weight"), q("norm2.bias"), 1e-5)
            a = torch.nn.functional.mish(h2 @ q("linear1.weight").t() + q("linear1.bias")) * m2
            X = X + (a @ q("linear2.weight").t() + q("linear2.bias")) * m3
        logits.append((X[0] * mh[b]) @ P["cp_ffn.1.weight"].view(-1) + P["cp_ffn.1.bias"].view(()))
    ref_logits = torch.stack(logits)
    (ref_logits * up.double()).sum().backward()
    assert np.abs(y.detach().cpu().numpy().ravel() - ref_logits.detach().numpy()).max() <= 3e-3 * max(1.0, float(ref_logits.abs().max()))
    bad = {}
    for k, g in ours.items():
        e = nrm((g - P[k].grad).numpy()) / max(nrm(P[k].grad.numpy()), 1e-30)
        if not e <= 5e-3:
            bad[k] = e
    assert not bad, bad
    assert set(ours) == {k for k, v in P.items() if v.grad is not None}


def test_indexed_training_step_equals_the_padded_one():
    """N3 x N1: a training step fed by (item_index, cu_seqlens) into the resident table == the padded-tensor step."""
    from src.models.datatypes import OutfitCompatibilityPredictionTask as CP
    n_items = [5, 16, 1, 8, 3, 0]
    g = np.random.default_rng(8)
    table = synth.item_embeddings(8, "table", 300)
    idx = [g.integers(0, 300, n) for n in n_items]
    cu_items = torch.from_numpy(np.concatenate([[0], np.cumsum(n_items)]).astype(np.int32))
    emb = np.zeros((len(n_items), 16, 1024), np.float32); mask = np.ones((len(n_items), 16), bool)
    for b, r in enumerate(idx):
        emb[b, :len(r)] = table[r]; mask[b, :len(r)] = False
    m = make_model("bf16")
    m.set_embedding_table(torch.from_numpy(table))
    up = torch.linspace(-1.0, 2.0, len(n_items)).cuda()
    (m(task=CP, outfit_embedding=cu(emb), outfit_mask=cu(mask)).squeeze(-1) * up).sum().backward()
    dense = {k: v.grad.clone() for k, v in trainable(m).items() if v.grad is not None}
    m.zero_grad(set_to_none=True)
    y = m(task=CP, item_index=torch.from_numpy(np.concatenate(idx).astype(np.int32)), cu_seqlens=cu_items)
    (y.squeeze(-1) * up).sum().backward()
    for k, v in trainable(m).items():
        if k in dense:
            assert torch.equal(v.grad, dense[k]), k


def test_training_guards():
    from src.models.datatypes import (OutfitCompatibilityPredictionTask as CP, OutfitComplementaryItemRetrievalTask as CIR)
    emb, mask = synth.outfit_batch(5, 2, 4, 3)
    m = make_model("bf16", dropout=0.3)
    with torch.no_grad():                      # scoring in train() mode under no_grad stays available (and has no dropout)
        a = m(task=CP, outfit_embedding=cu(emb), outfit_mask=cu(mask))
        b = m(task=CP, outfit_embedding=cu(emb), outfit_mask=cu(mask))
    assert torch.equal(a, b)
    m.cfg.transformer.dropout = 0.0
    with pytest.raises(NotImplementedError):        # encoder fine-tuning (gradients into the embeddings) is not built
        m(task=CP, outfit_embedding=cu(emb).requires_grad_(True), outfit_mask=cu(mask))
    m.train_precision = "bf16x3"
    with pytest.raises(ValueError):
        m(task=CP, outfit_embedding=cu(emb), outfit_mask=cu(mask))


@pytest.mark.parametrize("prec,gtol,ltol", [("f16", 5e-3, 2e-3), ("bf16", 3e-2, 1e-2)])
def test_cir_train_step_vs_reference_golden(prec, gtol, ltol):
    """CIR trainer step (complementary_item_retrieval_trainer.py:73-92): y_hat = model(CIR batch) in train() mode,
    SetWiseRankingLoss(margin 2), backward — against the reference's own loss / y_hat / gradients (dropout 0)."""
    from src.losses import SetWiseRankingLoss
    from src.models.datatypes import OutfitComplementaryItemRetrievalTask as CIR
    g = golden("train_step_cir")
    n, seed, K = g["n_items"], int(g["seed"]), int(g["K"])
    B = len(n)
    emb, mask = synth.outfit_batch(seed, B, 16, n)
    assert synth.checksum(emb) == str(g["emb_crc"])
    txt = synth.unit_rows(seed, "target_text", B, 512)
    pos = synth.item_embeddings(seed, "pos", B) * 3.0
    neg = synth.item_embeddings(seed, "neg", B * K).reshape(B, K, 1024) * 3.0
    m = make_model(prec)
    y = m(task=CIR, outfit_embedding=cu(emb), outfit_mask=cu(mask), target_item_text_embedding=cu(txt))
    loss = SetWiseRankingLoss(margin=2.0)(batch_y=cu(pos), batch_y_hat=y, batch_negative_samples=cu(neg), batch_negative_mask=cu(g["neg_mask"]))
    loss.backward()
    assert np.abs(y.detach().cpu().numpy() - g["y_hat"]).max() <= ltol * np.abs(g["y_hat"]).max()
    assert abs(float(loss) - float(g["loss"])) <= ltol * abs(float(g["loss"]))
    params = trainable(m)
    for k in g["no_grad_names"]:
        assert params[str(k)].grad is None, k
    bad = {}
    for k, ref_norm in zip(g["grad_names"], g["grad_norms"]):
        k = str(k)
        gr = params[k].grad.detach().cpu().numpy()
        if "grad/" + k in g.files:
            err = nrm(gr - g["grad/" + k]) / max(ref_norm, 1e-30)
        else:
            ref = g["gsample/" + k]
            err = nrm(gr.ravel()[::1009] - ref) / max(nrm(ref), 1e-30)
            assert abs(nrm(gr) - ref_norm) <= gtol * ref_norm, (k, nrm(gr), ref_norm)
        if not err <= gtol:
            bad[k] = err
    assert not bad, bad


def test_cir_training_with_dropout_and_indexed_input_runs_and_is_reproducible():
    from src.models.datatypes import OutfitFillInTheBlankTask as FITB
    n_items = [4, 9, 1]
    g = np.random.default_rng(3)
    table = synth.item_embeddings(3, "table", 64)
    idx = torch.from_numpy(np.concatenate([g.integers(0, 64, n) for n in n_items]).astype(np.int32))
    cu_items = torch.from_numpy(np.concatenate([[0], np.cumsum(n_items)]).astype(np.int32))
    txt = cu(synth.unit_rows(3, "t", 3, 512))
    m = make_model("bf16", dropout=0.3)
    m.set_embedding_table(torch.from_numpy(table))
    outs = []
    for _ in range(2):
        torch.manual_seed(5)
        m.zero_grad(set_to_none=True)
        y = m(task=FITB, item_index=idx, cu_seqlens=cu_items, target_item_text_embedding=txt)
        y.square().sum().backward()
        outs.append((y.detach().clone(), m.cir_ffn[0].weight.grad.clone(), m.target_item_image_emb.grad.clone()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    assert m.outfit_token.grad is None and m.cp_ffn[1].weight.grad is None
    m.eval()
    with torch.no_grad():
        y_eval = m(task=FITB, item_index=idx, cu_seqlens=cu_items, target_item_text_embedding=txt)
    assert not torch.equal(y_eval, outs[0][0])          # dropout was really applied in train mode


def test_training_step_properties_at_config5_size():
    """Size-independent properties at BASELINE config 5's per-GPU batch (256 outfits x 8 items), where no CPU reference is
    affordable: (1) the backward is linear in the upstream gradient - doubling d loss / d logits doubles every parameter gradient
    EXACTLY (powers of two commute with every rounding); (2) gradients of a batch = sum of the gradients of its two halves
    (outfits are independent), to fp32 accumulation-order rounding - checked with split-K off, because a different K
    partition re-associates the fp32 sums, which flips a few bf16 roundings of the activations, and single-product bf16 then
    differs at its own 1e-2 noise level between batch compositions (round-1 scratch script, git 81b82dd); (3) two runs are bit-identical
    (deterministic reductions)."""
    from outfitx_amd import _lib as L
    from src.models.datatypes import OutfitCompatibilityPredictionTask as CP
    B = 256
    emb, mask = synth.outfit_batch(2024, B, 16, synth.ragged_lengths(2024, B, 1, 16))
    m = make_model("bf16")
    x, k = cu(emb), cu(mask)
    up = torch.linspace(-1.0, 1.0, B).cuda()

    def grads(sl, scale):
        m.zero_grad(set_to_none=True)
        y = m(task=CP, outfit_embedding=x[sl], outfit_mask=k[sl]).squeeze(-1)
        (y * up[sl] * scale).sum().backward()
        return {n: p.grad.clone() for n, p in trainable(m).items() if p.grad is not None}

    full, full2, again = grads(slice(0, B), 1.0), grads(slice(0, B), 2.0), grads(slice(0, B), 1.0)
    for n in full:
        assert torch.equal(full[n], again[n]), n
        assert torch.equal(full[n] * 2, full2[n]), n
    L.load().ofx_tune(5, 0)
    try:
        full = grads(slice(0, B), 1.0)
        lo, hi = grads(slice(0, B // 2), 1.0), grads(slice(B // 2, B), 1.0)
    finally:
        L.load().ofx_tune(5, 1)
    for n in full:
        ref = full[n].double()
        if float(ref.norm()) < 1e-6:
            continue                    # sum_b up_b = 0 makes the last layer's linear2 bias gradient vanish identically
        err = float((lo[n].double() + hi[n].double() - ref).norm() / ref.norm())
        assert err < 1e-4, (n, err)


@pytest.mark.parametrize("task", ["cp", "cir"])
def test_gradient_sink_equals_autograd_accumulation(task):
    """trainer.FlatGrads + OutfitX.grad_sink: the backward kernels add straight into the arena views (parameter-shaped
    destinations, real FFN extents, C += result in the GEMM / reduce epilogues) - bit-identical to letting autograd accumulate
    the returned gradients, over two accumulated micro-batches with different batch shapes."""
    from outfitx_amd.trainer import FlatGrads
    from src.models.datatypes import OutfitCompatibilityPredictionTask as CP, OutfitComplementaryItemRetrievalTask as CIR
    batches = []
    for seed, n in ((61, [3, 9, 1, 16, 5]), (62, [8, 8, 2])):
        emb, mask = synth.outfit_batch(seed, len(n), 16, np.array(n))
        batches.append((cu(emb), cu(mask), cu(synth.unit_rows(seed, "t", len(n), 512)), torch.linspace(-1.0, 2.0, len(n)).cuda()))
    results = []
    for sink in (True, False):
        m = make_model("bf16")
        params = list(trainable(m).values())
        if sink:
            fg = FlatGrads(params)
            m.grad_sink = True
        for emb, mask, txt, up in batches:
            if task == "cp":
                (m(task=CP, outfit_embedding=emb, outfit_mask=mask).squeeze(-1) * up).sum().backward()
            else:
                (m(task=CIR, outfit_embedding=emb, outfit_mask=mask, target_item_text_embedding=txt) * up[:, None]).sum().backward()
        results.append({k: (None if p.grad is None else p.grad.clone()) for k, p in trainable(m).items()})
        if sink:
            on_path = [k for k, p in trainable(m).items() if p.grad is not None and p.grad.abs().sum() > 0]
            assert all(p.grad.data_ptr() >= fg.flat.data_ptr() for p in params)          # still the arena views
    a, b = results
    off_path = {"cp": ("target_item_image_emb", "cir_ffn.0.weight"), "cir": ("outfit_token", "cp_ffn.1.weight", "cp_ffn.1.bias")}[task]
    for k in b:
        if k in off_path:
            assert b[k] is None and not a[k].any(), k                                    # untouched zeros in the arena
        else:
            assert torch.equal(a[k], b[k]), k
    assert len(on_path) == len(b) - len(off_path)


def test_trainer_overfits_a_small_batch():
    """End-to-end sanity of the whole loop (tape forward, fused focal loss, sink backward, clip, fused AdamW, OneCycleLR, re-pack):
    32 outfits with random labels are memorised - the focal loss falls by more than 10x and every outfit is classified."""
    from outfitx_amd.trainer import CPTrainConfig, CPTrainer
    from src.models.datatypes import OutfitCompatibilityPredictionTask as CP
    emb, mask = synth.outfit_batch(77, 32, 16, synth.ragged_lengths(77, 32, 2, 12))
    lab = torch.from_numpy((np.random.default_rng(7).random(32) < 0.5).astype(np.float32))
    batch = {"input_dict": {"task": CP, "outfit_embedding": torch.from_numpy(emb), "outfit_mask": torch.from_numpy(mask)}, "label": lab}
    m = make_model("bf16")
    tr = CPTrainer(m, steps_per_epoch=120, cfg=CPTrainConfig(learning_rate=3e-4, accumulation_steps=1, n_epochs=1), params=list(trainable(m).values()))
    losses = [float(tr.micro_step(batch, i)[0]) for i in range(120)]
    m.eval()
    with torch.no_grad():
        y = m(task=CP, outfit_embedding=cu(emb), outfit_mask=cu(mask)).squeeze(-1).cpu()
    print("overfit: loss %.4f -> %.6f, correct %d / 32" % (losses[0], losses[-1], int(((y > 0).float() == lab).sum())))
    assert losses[-1] < 0.2 * losses[0], (losses[0], losses[-1])
    assert int(((y > 0).float() == lab).sum()) >= 30


def test_fused_optimizer_steps_are_noticed():
    """torch.optim.AdamW(fused=True) rewrites the parameters without bumping their version counters; the global optimizer
    post-step hook must still make the next forward re-pack the operand copies (reference-style loop, no CPTrainer)."""
    from src.models.datatypes import OutfitCompatibilityPredictionTask as CP
    emb, mask = synth.outfit_batch(12, 6, 16, synth.ragged_lengths(12, 6, 2, 9))
    m = make_model("bf16")
    opt = torch.optim.AdamW(list(trainable(m).values()), lr=1e-3, fused=True)
    x, k = cu(emb), cu(mask)
    y0 = m(task=CP, outfit_embedding=x, outfit_mask=k)
    y0.sum().backward()
    v0 = m.cp_ffn[1].weight._version
    opt.step()
    y1 = m(task=CP, outfit_embedding=x, outfit_mask=k)
    assert not torch.equal(y0, y1), "stale packed weights after a fused optimizer step"
    m.eval()
    with torch.no_grad():
        y2 = m(task=CP, outfit_embedding=x, outfit_mask=k)          # the scoring engine (bf16x3 copies) is re-packed too
    assert float((y2 - y1).abs().max()) < 0.05 * float(y1.abs().max()) + 0.05


def test_training_step_under_distributed_data_parallel():
    """The reference wraps the model in DistributedDataParallel (distributed_trainer.py:315-329).  Our hand-written backward
    reaches DDP's gradient hooks like any autograd node: a world-size-1 RCCL group, DDP(find_unused_parameters=True) as the
    reference configures it, one step - gradients equal the unwrapped model's."""
    import socket
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from src.models.datatypes import OutfitCompatibilityPredictionTask as CP
    emb, mask = synth.outfit_batch(31, 6, 16, synth.ragged_lengths(31, 6, 1, 10))
    up = torch.linspace(-1.0, 2.0, 6).cuda()
    plain = make_model("bf16")
    (plain(task=CP, outfit_embedding=cu(emb), outfit_mask=cu(mask)).squeeze(-1) * up).sum().backward()
    want = {k: v.grad.clone() for k, v in trainable(plain).items() if v.grad is not None}
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        m = make_model("bf16")
        ddp = DDP(m, device_ids=[0], find_unused_parameters=True, broadcast_buffers=True)
        (ddp(task=CP, outfit_embedding=cu(emb), outfit_mask=cu(mask)).squeeze(-1) * up).sum().backward()
        got = {k: v.grad.clone() for k, v in trainable(m).items() if v.grad is not None}
    finally:
        dist.destroy_process_group()
    assert set(got) >= set(want)
    for k in want:
        assert torch.equal(got[k], want[k]), k


def test_overlapped_gradient_reduction_equals_the_plain_one():
    """trainer.CPTrainer with more than one rank reduces layer l's gradient slice on a side stream behind the HIP event that the
    backward records when that layer is final (ofx_train_arm_layer_events).  Here on ONE device: a world-size-1 RCCL group with the
    trainer told the world is 2 (so every slice goes through an async all-reduce on the comm stream and is then halved) - the
    parameters after three optimizer steps must be bit-identical to the slice-by-slice reduction after the backward, every armed
    event must have fired, and an armed backward must leave the gradients themselves unchanged."""
    import socket
    import torch.distributed as dist
    from outfitx_amd.trainer import CPTrainConfig, CPTrainer
    from src.models.datatypes import OutfitCompatibilityPredictionTask as CP
    emb, mask = synth.outfit_batch(91, 24, 16, synth.ragged_lengths(91, 24, 1, 12))
    lab = torch.from_numpy((np.random.default_rng(9).random(24) < 0.5).astype(np.float32))
    batch = {"input_dict": {"task": CP, "outfit_embedding": torch.from_numpy(emb), "outfit_mask": torch.from_numpy(mask)}, "label": lab}
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        finals = []
        for overlap in (True, False):
            m = make_model("bf16")
            tr = CPTrainer(m, steps_per_epoch=6, cfg=CPTrainConfig(learning_rate=1e-3, accumulation_steps=2, n_epochs=1, fused_optimizer=False),
                           params=list(trainable(m).values()))
            assert tr.layer_slices is not None and len(tr.layer_slices) == 6
            tr.overlap_reduce = overlap
            tr._world = lambda: 2
            for i in range(6):
                tr.micro_step(batch, i)
            torch.cuda.synchronize()
            if overlap:
                assert tr._layer_events is not None and all(e.query() for e in tr._layer_events)
            else:
                assert tr._layer_events is None
            finals.append({k: v.detach().clone() for k, v in trainable(m).items()})
        for k in finals[0]:
            assert torch.equal(finals[0][k], finals[1][k]), k
    finally:
        dist.destroy_process_group()


def test_overlapped_reduction_is_not_armed_when_a_gradient_is_detached_from_the_arena():
    """The layer events mean "final in the arena" only when the backward adds straight into the parameters' own .grad views
    (gradient-sink path).  With one .grad replaced by a non-contiguous tensor the backward goes through autograd's accumulation,
    which runs after the events: CPTrainer must then refuse to arm (plain reduction after the backward)."""
    from outfitx_amd.trainer import CPTrainConfig, CPTrainer
    m = make_model("bf16")
    tr = CPTrainer(m, steps_per_epoch=2, cfg=CPTrainConfig(learning_rate=1e-3, accumulation_steps=1, n_epochs=1, fused_optimizer=False),
                   params=list(trainable(m).values()))
    tr._world = lambda: 2
    assert m.sink_ready() and tr._arm_overlap()
    w = m.transformer_encoder.layers[2].linear1.weight
    w.grad = torch.zeros(w.shape[1], w.shape[0], device=w.device).t()        # same shape, not contiguous: no longer an arena view
    assert not m.sink_ready() and not tr._arm_overlap()


def test_rccl_packed_record_allgather_world1():
    """parallel.sharded_topk's ONE collective - all_gather_into_tensor of the packed (global index int64 | distance fp32) byte record -
    on RCCL itself (a world-size-1 group on this device: the byte dtype and the flat-output form are what must be accepted), and the
    unpacking round trip."""
    import socket
    import torch.distributed as dist
    nq, k = 37, 50
    idx = torch.randint(0, 100000, (nq, k), dtype=torch.int64, device="cuda")
    dst = torch.rand(nq, k, device="cuda")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        nb_i, nb_d = idx.numel() * 8, dst.numel() * 4
        rec = torch.empty(nb_i + nb_d, dtype=torch.uint8, device="cuda")
        rec[:nb_i] = idx.view(torch.uint8).reshape(-1); rec[nb_i:] = dst.view(torch.uint8).reshape(-1)
        flat = torch.empty(nb_i + nb_d, dtype=torch.uint8, device="cuda")
        dist.all_gather_into_tensor(flat, rec)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    allrec = flat.view(1, -1)
    assert torch.equal(allrec[:, :nb_i].contiguous().view(torch.int64).view(1, nq, k)[0], idx)
    assert torch.equal(allrec[:, nb_i:].contiguous().view(torch.float32).view(1, nq, k)[0], dst)


def test_reference_trainer_step_verbatim():
    """The reference's train_epoch body (compatibility_prediction_trainer.py:57-81) line for line against our model: autocast,
    FocalLoss, /accumulation, GradScaler.scale(loss).backward(), unscale_, clip_grad_norm_, scaler.step, update, zero_grad,
    OneCycleLR - two optimizer steps with accumulation 2 and dropout 0.3; the loss stays finite and the weights move."""
    from torch.amp import GradScaler, autocast
    from src.losses import FocalLoss
    from src.models.datatypes import OutfitCompatibilityPredictionTask as CP
    model = make_model("bf16", dropout=0.3)
    params = [p for p in model.parameters() if p.requires_grad]
    optimizer = torch.optim.AdamW(model.parameters(), lr=2e-5)
    scheduler = torch.optim.lr_scheduler.OneCycleLR(optimizer=optimizer, max_lr=2e-5, epochs=1, steps_per_epoch=2, pct_start=0.3,
                                                    anneal_strategy="cos", div_factor=25, final_div_factor=1e4)
    scaler, loss_fn, accumulation_steps = GradScaler(), FocalLoss(alpha=0.75, gamma=2, reduction="mean"), 2
    w0 = model.transformer_encoder.layers[0].linear1.weight.detach().clone()
    optimizer.zero_grad()
    losses = []
    for step in range(4):
        emb, mask = synth.outfit_batch(100 + step, 16, 16, synth.ragged_lengths(100 + step, 16, 1, 12))
        input_dict = {"task": CP, "outfit_embedding": torch.from_numpy(emb).to(0), "outfit_mask": torch.from_numpy(mask).to(0)}
        with autocast(enabled=True, device_type="cuda"):
            y_hats = model(**input_dict).squeeze(dim=-1)
            labels = (torch.arange(16) % 2).float().to(0)
            loss = loss_fn(y_hat=y_hats, y_true=labels)
            original_loss = loss.clone().detach()
            loss = loss / accumulation_steps
        scaler.scale(loss).backward()
        if (step + 1) % accumulation_steps == 0:
            scaler.unscale_(optimizer)
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
            scaler.step(optimizer)
            scaler.update()
            optimizer.zero_grad()
            scheduler.step()
        losses.append(float(original_loss))
    assert all(np.isfinite(losses)) and len(params) == 77          # 75 on the CP path + target_item_image_emb + cir_ffn (no gradient here)
    assert not torch.equal(model.transformer_encoder.layers[0].linear1.weight.detach(), w0)
