"""CPU oracle for the OutfitX compatibility-scoring forward path — TEST INFRASTRUCTURE ONLY.

A plain-numpy restatement of the arithmetic the reference executes on this path.  It is the
checker for the HIP kernels; nothing under outfitx_amd/ may import it.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg use it.

Parity status: PINNED.  The reference's own tests hold no numerical vectors for this path
(SURVEY.md §4), so this restatement is pinned against outputs of the reference itself, run in the
build container by oracle/gen_golden.py (which imports /root/reference's src.models) and committed
as tests/golden/*.npz; tests/test_oracle_golden.py checks every function below against them.

The arithmetic that the reference delegates to third-party libraries is restated from the
published algorithms, anchored on the reference's call sites:
  * torch==2.5.0 (environment.yml:9): nn.TransformerEncoderLayer(norm_first, activation=F.mish),
    nn.MultiheadAttention, F.scaled_dot_product_attention, F.layer_norm, F.normalize, torch.cdist,
    torch.topk — call sites src/models/outfit_x.py:32-45,137-143,165-171.
  * transformers==4.48.3 (environment.yml:13): CLIPVisionModelWithProjection /
    CLIPTextModelWithProjection — call sites clip_image_encoder.py:74-76, clip_text_encoder.py:56-58.

All functions take and return numpy arrays; `W` is a dict name → ndarray with the reference's
state_dict key names.  `dt` selects the compute dtype (float32 = what the
reference does; float64 = a tighter "truth" for error budgeting).
"""
from __future__ import annotations

import numpy as np

LN_EPS = 1e-5


# ----------------------------------------------------------------- primitives
def layer_norm(x, w, b, eps=LN_EPS):
    """F.layer_norm over the last axis, biased variance (torch default; outfit_x.py:32-40)."""
    mu = x.mean(-1, keepdims=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdims=True)
    return xc / np.sqrt(var + x.dtype.type(eps)) * w + b


def softplus(u):
    """torch softplus, beta=1, threshold=20."""
    return np.where(u > 20, u, np.log1p(np.exp(np.minimum(u, 20))))


def mish(u):
    """F.mish = u * tanh(softplus(u))  (transformer_config.py:22)."""
    return u * np.tanh(softplus(u))


def quick_gelu(u):
    """HF ACT2FN['quick_gelu'] = u * sigmoid(1.702 u) (CLIP default hidden_act)."""
    return u / (1.0 + np.exp(-u.dtype.type(1.702) * u))


def gelu(u):
    from math import sqrt
    try:
        from scipy.special import erf
    except Exception:  # pragma: no cover
        erf = np.vectorize(__import__("math").erf)
    return 0.5 * u * (1.0 + erf(u / u.dtype.type(sqrt(2.0))))


ACTS = {"mish": mish, "quick_gelu": quick_gelu, "gelu": gelu}


def softmax_masked(s, key_dead):
    """softmax over the last axis with -inf on dead keys.  key_dead broadcasts to s."""
    s = np.where(key_dead, -np.inf, s)
    m = s.max(-1, keepdims=True)
    e = np.exp(s - m)
    return e / e.sum(-1, keepdims=True)


def l2_normalize(x, eps=1e-12):
    """F.normalize(p=2, dim=-1) (base_image_encoder.py:46-47, base_text_encoder.py:37-38)."""
    n = np.sqrt((x * x).sum(-1, keepdims=True))
    return x / np.maximum(n, x.dtype.type(eps))


def _mha(h, dead, Wq, bq, Wk, bk, Wv, bv, Wo, bo, n_head, extra_dead=None):
    """Multi-head attention: h [B,S,D]; dead [B,S] True = key ignored; weights [out,in]."""
    B, S, D = h.shape
    dh = D // n_head
    q = (h @ Wq.T + bq).reshape(B, S, n_head, dh).transpose(0, 2, 1, 3)
    k = (h @ Wk.T + bk).reshape(B, S, n_head, dh).transpose(0, 2, 1, 3)
    v = (h @ Wv.T + bv).reshape(B, S, n_head, dh).transpose(0, 2, 1, 3)
    s = (q @ k.transpose(0, 1, 3, 2)) * h.dtype.type(dh ** -0.5)
    kd = dead[:, None, None, :]
    if extra_dead is not None:
        kd = kd | extra_dead[None, None, :, :]
    a = softmax_masked(s, kd)
    o = (a @ v).transpose(0, 2, 1, 3).reshape(B, S, D)
    return o @ Wo.T + bo


# ------------------------------------------------- row D: global outfit Transformer
def outfit_encoder(x, pad, W, dt=np.float32, n_layers=6, n_head=16, taps=None):
    """6 × pre-norm TransformerEncoderLayer(d=1024, 16 heads, ffn 2024, mish), eval mode, no
    final norm (built at outfit_x.py:32-45, called :137-140,:165-168).
    x [B,S,D]; pad [B,S] bool, True = padded key.  `taps`, if a list, receives X[:,0] per layer."""
    x = x.astype(dt)
    D = x.shape[-1]
    for i in range(n_layers):
        p = f"transformer_encoder.layers.{i}."
        g = lambda k: W[p + k].astype(dt)
        Win, bin_ = g("self_attn.in_proj_weight"), g("self_attn.in_proj_bias")
        h = layer_norm(x, g("norm1.weight"), g("norm1.bias"))
        x = x + _mha(h, pad,
                     Win[:D], bin_[:D], Win[D:2 * D], bin_[D:2 * D], Win[2 * D:], bin_[2 * D:],
                     g("self_attn.out_proj.weight"), g("self_attn.out_proj.bias"), n_head)
        h = layer_norm(x, g("norm2.weight"), g("norm2.bias"))
        u = h @ g("linear1.weight").T + g("linear1.bias")
        x = x + mish(u) @ g("linear2.weight").T + g("linear2.bias")
        if taps is not None:
            taps.append(x[:, 0].copy())
    return x


def _prefix(tok, emb, mask):
    B, _, D = emb.shape
    tok = np.broadcast_to(tok.reshape(-1, D), (B, D))[:, None, :]      # [D] or [B,D] → [B,1,D]
    x = np.concatenate([tok, emb], 1)
    pad = np.concatenate([np.zeros((B, 1), bool), mask.astype(bool)], 1)
    return x, pad


# ------------------------------------------------- row B: OutfitX._cp_forward
def cp_forward(outfit_embedding, outfit_mask, W, dt=np.float32, taps=None):
    """outfit_x.py:120-144: [outfit_token ; items] → encoder → row 0 → Linear(1024,1) → [B,1]
    raw logit (Dropout = identity in eval; no sigmoid)."""
    emb = outfit_embedding.astype(dt)
    x, pad = _prefix(W["outfit_token"].astype(dt), emb, outfit_mask)
    y = outfit_encoder(x, pad, W, dt, taps=taps)[:, 0]
    return y @ W["cp_ffn.1.weight"].astype(dt).T + W["cp_ffn.1.bias"].astype(dt)


# ------------------------------------------------- row C: OutfitX._cir_forward
def cir_forward(outfit_embedding, outfit_mask, target_item_text_embedding, W, dt=np.float32, taps=None):
    """outfit_x.py:147-172: prefix token = [target_item_image_emb ‖ target text emb] per outfit;
    encoder; row 0 → Linear(1024,1024, bias=False) → [B,1024]."""
    emb = outfit_embedding.astype(dt)
    B = emb.shape[0]
    tok = np.concatenate([np.broadcast_to(W["target_item_image_emb"].astype(dt), (B, 512)),
                          target_item_text_embedding.astype(dt)], -1)
    x, pad = _prefix(tok, emb, outfit_mask)
    y = outfit_encoder(x, pad, W, dt, taps=taps)[:, 0]
    return y @ W["cir_ffn.0.weight"].astype(dt).T


# ------------------------------------------------- rows F/G: CLIP towers
def _clip_layers(x, dead, W, prefix, n_layers, n_head, act, dt, extra_dead=None):
    f = ACTS[act]
    for i in range(n_layers):
        p = f"{prefix}encoder.layers.{i}."
        g = lambda k: W[p + k].astype(dt)
        h = layer_norm(x, g("layer_norm1.weight"), g("layer_norm1.bias"))
        x = x + _mha(h, dead,
                     g("self_attn.q_proj.weight"), g("self_attn.q_proj.bias"),
                     g("self_attn.k_proj.weight"), g("self_attn.k_proj.bias"),
                     g("self_attn.v_proj.weight"), g("self_attn.v_proj.bias"),
                     g("self_attn.out_proj.weight"), g("self_attn.out_proj.bias"), n_head, extra_dead)
        h = layer_norm(x, g("layer_norm2.weight"), g("layer_norm2.bias"))
        x = x + f(h @ g("mlp.fc1.weight").T + g("mlp.fc1.bias")) @ g("mlp.fc2.weight").T + g("mlp.fc2.bias")
    return x


def patchify(pixels, patch=32):
    """[N,3,H,W] → [N, (H/p)(W/p), 3·p·p]: row = patch (row-major py,px), col = (c,ky,kx) — the
    im2col of Conv2d(3,768,k=32,s=32,bias=False) followed by flatten(2).transpose(1,2)."""
    N, C, H, Wd = pixels.shape
    gy, gx = H // patch, Wd // patch
    p = pixels.reshape(N, C, gy, patch, gx, patch).transpose(0, 2, 4, 1, 3, 5)
    return p.reshape(N, gy * gx, C * patch * patch)


def vit_forward(pixels, W, dt=np.float32, act="quick_gelu", n_layers=12, n_head=12, prefix="vision_model."):
    """HF CLIPVisionModelWithProjection(pixel_values).image_embeds (clip_image_encoder.py:74-76):
    patch conv → [CLS ; patches] + pos → pre_layrnorm → 12 pre-LN blocks → post_layernorm(CLS)
    → visual_projection (no bias).  Returns UN-normalised [N,512]."""
    g = lambda k: W[k].astype(dt)
    pw = g(prefix + "embeddings.patch_embedding.weight")
    width = pw.shape[0]
    pe = patchify(pixels.astype(dt), pw.shape[-1]) @ pw.reshape(width, -1).T
    N = pe.shape[0]
    cls = np.broadcast_to(g(prefix + "embeddings.class_embedding"), (N, 1, width))
    x = np.concatenate([cls, pe], 1) + g(prefix + "embeddings.position_embedding.weight")[None]
    x = layer_norm(x, g(prefix + "pre_layrnorm.weight"), g(prefix + "pre_layrnorm.bias"))
    dead = np.zeros(x.shape[:2], bool)
    x = _clip_layers(x, dead, W, prefix, n_layers, n_head, act, dt)
    pooled = layer_norm(x[:, 0], g(prefix + "post_layernorm.weight"), g(prefix + "post_layernorm.bias"))
    return pooled @ g("visual_projection.weight").T


def eos_positions(input_ids, eos_token_id=49407):
    """HF CLIPTextModel pooling index: legacy eos_token_id==2 → argmax(ids); else first == eos."""
    if eos_token_id == 2:
        return input_ids.argmax(-1)
    return (input_ids == eos_token_id).astype(np.int32).argmax(-1)


def text_forward(input_ids, attention_mask, W, dt=np.float32, act="quick_gelu", n_layers=12, n_head=8,
                 eos_token_id=49407, prefix="text_model."):
    """HF CLIPTextModelWithProjection(input_ids, attention_mask).text_embeds
    (clip_text_encoder.py:56-58): tok+pos emb → 12 pre-LN blocks under causal ∧ key-padding mask →
    final_layer_norm → hidden state at the EOS position → text_projection.  UN-normalised [N,512]."""
    g = lambda k: W[k].astype(dt)
    N, T = input_ids.shape
    x = g(prefix + "embeddings.token_embedding.weight")[input_ids] + \
        g(prefix + "embeddings.position_embedding.weight")[:T][None]
    dead = attention_mask == 0
    causal_dead = np.triu(np.ones((T, T), bool), 1)      # key j > query i
    x = _clip_layers(x, dead, W, prefix, n_layers, n_head, act, dt, extra_dead=causal_dead)
    x = layer_norm(x, g(prefix + "final_layer_norm.weight"), g(prefix + "final_layer_norm.bias"))
    pooled = x[np.arange(N), eos_positions(input_ids, eos_token_id)]
    return pooled @ g("text_projection.weight").T


# ------------------------------------------------- row E: ItemEncoder.forward / fuser
def aggregate_embeddings(img, txt, method="concat"):
    """src/utils/model_utils.py:26-45.  'mean' reproduces the reference's literal tensor
    semantics, mean(stack([img,txt]), dim=-2) → [2,B,512] (shape-buggy upstream)."""
    if method == "concat":
        return np.concatenate([img, txt], -1)
    if method == "mean":
        return np.stack([img, txt]).mean(-2)
    raise ValueError(f"Unsupported aggregation method: {method}. Use 'concat' or 'mean'.")


def item_encoder(pixels, input_ids, attention_mask, W_img, W_txt, dt=np.float32, norm_out=True,
                 method="concat", **kw):
    """item_encoder.py:46-61 on tensor inputs: pixels [B,L,3,224,224], ids/mask [B,L,T] → [B,L,1024]."""
    B, L = pixels.shape[:2]
    img = vit_forward(pixels.reshape(B * L, *pixels.shape[2:]), W_img, dt, **kw).reshape(B, L, -1)
    T = input_ids.shape[-1]
    txt = text_forward(input_ids.reshape(B * L, T), attention_mask.reshape(B * L, T), W_txt, dt, **kw).reshape(B, L, -1)
    if norm_out:
        img, txt = l2_normalize(img), l2_normalize(txt)
    return aggregate_embeddings(img, txt, method)


# ------------------------------------------------- rows H/I: FITB / CIR scoring
def fitb_argmin(y_hat, candidates):
    """fill_in_the_blank_trainer.py:50-56: cdist(y[B,1,D], cand[B,C,D]).squeeze(1).argmin(-1);
    torch.cdist uses the direct sqrt(sum((a-b)^2)) form for ≤25 rows.  Returns (idx int64, dist)."""
    d = np.sqrt(((y_hat[:, None, :].astype(np.float32) - candidates.astype(np.float32)) ** 2).sum(-1, dtype=np.float32))
    return d.argmin(-1).astype(np.int64), d


def l2_topk(Q, P, k=50, chunk=256):
    """complementary_item_retrieval_trainer.py:240-249: cdist(Q,P) (‖q‖²+‖p‖²−2q·p form, fp32,
    clamped at 0, sqrt) → topk(k, largest=False): ascending, ties → smaller index first here.
    Returns (idx [nq,k] int64, dist [nq,k] float32)."""
    Q = Q.astype(np.float32); P = P.astype(np.float32)
    pn = (P * P).sum(-1)
    idx = np.empty((Q.shape[0], k), np.int64); dist = np.empty((Q.shape[0], k), np.float32)
    for s in range(0, Q.shape[0], chunk):
        q = Q[s:s + chunk]
        d2 = np.maximum((q * q).sum(-1)[:, None] + pn[None, :] - 2.0 * (q @ P.T), 0.0)
        d = np.sqrt(d2, dtype=np.float32)
        order = np.argsort(d, axis=-1, kind="stable")[:, :k]
        idx[s:s + chunk] = order
        dist[s:s + chunk] = np.take_along_axis(d, order, -1)
    return idx, dist


def l2_dist_exact(Q, P):
    """float64 direct-form distances, for tie analysis in tests."""
    Q = Q.astype(np.float64); P = P.astype(np.float64)
    return np.sqrt(np.maximum((Q * Q).sum(-1)[:, None] + (P * P).sum(-1)[None] - 2 * Q @ P.T, 0))


# ------------------------------------------------- next row N1: FocalLoss
def focal_loss(logits, y, alpha=0.75, gamma=2.0):
    """src/losses/focal_loss.py:23-41 (mean reduction)."""
    x = logits.astype(np.float64); y = y.astype(np.float64)
    ce = np.maximum(x, 0) - x * y + np.log1p(np.exp(-np.abs(x)))
    p = 1 / (1 + np.exp(-x))
    pt = p * y + (1 - p) * (1 - y)
    return float(((alpha * y + (1 - alpha) * (1 - y)) * ce * (1 - pt) ** gamma).mean())


# ------------------------------------------------- next row N3: collate padding
def pad_outfits(seqs, max_length=16, d=1024):
    """outfit_x_base_processor.py:20-81 with padding='max_length': truncate to max_length, pad
    with zero rows, mask True on pads.  seqs: list of [n_i, d] arrays."""
    B = len(seqs)
    emb = np.zeros((B, max_length, d), np.float32); mask = np.ones((B, max_length), bool)
    for i, s in enumerate(seqs):
        n = min(len(s), max_length)
        if n:
            emb[i, :n] = np.asarray(s, np.float32)[:n]
        mask[i, :n] = False
    return emb, mask


# ---------------------------------------------------------------------------------------------------------------------
# Image preprocessing ("next" row N2).  The reference calls transformers.CLIPImageProcessor(do_convert_rgb=False) on the host
# (src/models/encoders/image_encoders/clip_image_encoder.py:29-31,69-71); its resize is PIL's `Image.resize(BICUBIC)`, i.e.
# Pillow's libImaging/Resample.c (third-party; reference pins pillow==11.0.0 in environment.yml:12, the container has 12.2.0 —
# the routine is unchanged between them).  Restated here from its published algorithm and PINNED against the container's PIL
# itself (tests/test_oracle_golden.py::test_pil_bicubic_restatement_is_bit_exact).
_PBITS = 22


def _bicubic_filter(x):
    a = -0.5
    x = np.abs(x)
    r = np.zeros_like(x)
    m1 = x < 1.0
    m2 = (x >= 1.0) & (x < 2.0)
    r[m1] = ((a + 2.0) * x[m1] - (a + 3.0)) * x[m1] * x[m1] + 1
    r[m2] = (((x[m2] - 5) * x[m2] + 8) * x[m2] - 4) * a
    return r


def pil_bicubic_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc -> (kk int64 [out, ksize], bounds [out, 2] = (xmin, count))."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), np.int64)
    bounds = np.zeros((out_size, 2), np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size)
        n = xmax - xmin
        w = _bicubic_filter((np.arange(n, dtype=np.float64) + xmin - center + 0.5) * ss)
        ww = 0.0
        for v in w:                     # sequential sum, like the C loop
            ww += v
        if ww != 0.0:
            w = w / ww
        kk[xx, :n] = np.where(w < 0, (-0.5 + w * (1 << _PBITS)).astype(np.int64), (0.5 + w * (1 << _PBITS)).astype(np.int64))
        bounds[xx] = (xmin, n)
    return kk, bounds


def _resample_axis0(img, out_size):
    kk, b = pil_bicubic_coeffs(img.shape[0], out_size)
    out = np.zeros((out_size,) + img.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, n = b[xx]
        acc = (1 << (_PBITS - 1)) + (img[xmin:xmin + n].astype(np.int64) * kk[xx, :n].reshape((-1,) + (1,) * (img.ndim - 1))).sum(0)
        out[xx] = np.clip(acc >> _PBITS, 0, 255).astype(np.uint8)
    return out


def pil_resize_bicubic(img, nw, nh):
    """uint8 [H,W] or [H,W,C] -> [nh,nw(,C)]: horizontal pass, 8-bit intermediate, vertical pass (ImagingResample)."""
    o = img
    if nw != img.shape[1]:
        o = np.swapaxes(_resample_axis0(np.swapaxes(o, 0, 1), nw), 0, 1)
    if nh != img.shape[0]:
        o = _resample_axis0(o, nh)
    return np.ascontiguousarray(o)


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def clip_preprocess(images, size=224):
    """CLIPImageProcessor: shortest edge -> size (bicubic), centre crop, / 255, normalise -> [N,3,size,size] fp32."""
    out = np.empty((len(images), 3, size, size), np.float32)
    mean = np.asarray(CLIP_MEAN, np.float32).reshape(3, 1, 1)
    std = np.asarray(CLIP_STD, np.float32).reshape(3, 1, 1)
    for i, a in enumerate(images):
        a = np.asarray(a)
        h, w = a.shape[:2]
        short, long_ = (w, h) if w <= h else (h, w)
        new_long = int(size * long_ / short)
        nw, nh = (size, new_long) if w <= h else (new_long, size)
        r = pil_resize_bicubic(a, nw, nh)
        left, top = (nw - size) // 2, (nh - size) // 2
        r = r[top:top + size, left:left + size].astype(np.float32)
        if r.ndim == 2:
            r = np.repeat(r[:, :, None], 3, 2)
        out[i] = (r.transpose(2, 0, 1) * np.float32(1 / 255.0) - mean) / std
    return out
