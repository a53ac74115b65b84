"""Deterministic synthetic weights and inputs for the OutfitX scoring path.

Nothing here is model code: it is the seeded generator that the golden-vector
script (oracle/gen_golden.py), the parity tests and bench.py share, so that
fixtures only need to store OUTPUTS (the 202 M parameters and the pixel tensors
are regenerated from a seed on whichever box runs the test).

Every tensor is drawn from its own PCG64 stream keyed by (seed, crc32(name)), so
any subset of a state dict can be generated independently and in any order.

Shapes and key names follow the reference's state_dict as enumerated in
SURVEY.md §8(b) (reference: src/models/outfit_x.py:25-90 and the HF CLIP modules
built by src/models/encoders/image_encoders/clip_image_encoder.py:20-22 and
src/models/encoders/text_encoders/clip_text_encoder.py:19-21).
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np

# ---- dimensions of the path -------------------------------
D_MODEL = 1024
N_HEAD = 16
D_FFN = 2024
N_LAYERS = 6
D_HALF = 512

VIT_WIDTH, VIT_LAYERS, VIT_HEADS, VIT_MLP, VIT_PATCH, VIT_IMG, VIT_POS = 768, 12, 12, 3072, 32, 224, 50
TXT_WIDTH, TXT_LAYERS, TXT_HEADS, TXT_MLP, TXT_VOCAB, TXT_POS = 512, 12, 8, 2048, 49408, 77
PROJ_DIM = 512
BOS_ID, EOS_ID = 49406, 49407

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence([int(seed), zlib.crc32(name.encode())])))


def outfit_transformer_shapes() -> Dict[str, Tuple[int, ...]]:
    """Key → shape for the 51,155,313 trainable parameters (no item encoder)."""
    s: Dict[str, Tuple[int, ...]] = {
        "outfit_token": (D_MODEL,),
        "target_item_image_emb": (D_HALF,),
        "cp_ffn.1.weight": (1, D_MODEL),
        "cp_ffn.1.bias": (1,),
        "cir_ffn.0.weight": (D_MODEL, D_MODEL),
    }
    for i in range(N_LAYERS):
        p = f"transformer_encoder.layers.{i}."
        s[p + "self_attn.in_proj_weight"] = (3 * D_MODEL, D_MODEL)
        s[p + "self_attn.in_proj_bias"] = (3 * D_MODEL,)
        s[p + "self_attn.out_proj.weight"] = (D_MODEL, D_MODEL)
        s[p + "self_attn.out_proj.bias"] = (D_MODEL,)
        s[p + "linear1.weight"] = (D_FFN, D_MODEL)
        s[p + "linear1.bias"] = (D_FFN,)
        s[p + "linear2.weight"] = (D_MODEL, D_FFN)
        s[p + "linear2.bias"] = (D_MODEL,)
        for n in ("norm1", "norm2"):
            s[p + n + ".weight"] = (D_MODEL,)
            s[p + n + ".bias"] = (D_MODEL,)
    return s


def _clip_layer_shapes(prefix: str, width: int, mlp: int, n_layers: int) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    for i in range(n_layers):
        p = f"{prefix}encoder.layers.{i}."
        for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[p + f"self_attn.{proj}.weight"] = (width, width)
            s[p + f"self_attn.{proj}.bias"] = (width,)
        s[p + "layer_norm1.weight"] = (width,)
        s[p + "layer_norm1.bias"] = (width,)
        s[p + "mlp.fc1.weight"] = (mlp, width)
        s[p + "mlp.fc1.bias"] = (mlp,)
        s[p + "mlp.fc2.weight"] = (width, mlp)
        s[p + "mlp.fc2.bias"] = (width,)
        s[p + "layer_norm2.weight"] = (width,)
        s[p + "layer_norm2.bias"] = (width,)
    return s


def vision_shapes(n_layers: int = VIT_LAYERS) -> Dict[str, Tuple[int, ...]]:
    """Key → shape of HF CLIPVisionModelWithProjection (ViT-B/32), keys relative to `….image_enc.model.`"""
    v = "vision_model."
    s: Dict[str, Tuple[int, ...]] = {
        v + "embeddings.class_embedding": (VIT_WIDTH,),
        v + "embeddings.patch_embedding.weight": (VIT_WIDTH, 3, VIT_PATCH, VIT_PATCH),
        v + "embeddings.position_embedding.weight": (VIT_POS, VIT_WIDTH),
        v + "pre_layrnorm.weight": (VIT_WIDTH,),
        v + "pre_layrnorm.bias": (VIT_WIDTH,),
    }
    s.update(_clip_layer_shapes(v, VIT_WIDTH, VIT_MLP, n_layers))
    s[v + "post_layernorm.weight"] = (VIT_WIDTH,)
    s[v + "post_layernorm.bias"] = (VIT_WIDTH,)
    s["visual_projection.weight"] = (PROJ_DIM, VIT_WIDTH)
    return s


def text_shapes(n_layers: int = TXT_LAYERS) -> Dict[str, Tuple[int, ...]]:
    """Key → shape of HF CLIPTextModelWithProjection, keys relative to `….text_enc.model.`"""
    t = "text_model."
    s: Dict[str, Tuple[int, ...]] = {
        t + "embeddings.token_embedding.weight": (TXT_VOCAB, TXT_WIDTH),
        t + "embeddings.position_embedding.weight": (TXT_POS, TXT_WIDTH),
    }
    s.update(_clip_layer_shapes(t, TXT_WIDTH, TXT_MLP, n_layers))
    s[t + "final_layer_norm.weight"] = (TXT_WIDTH,)
    s[t + "final_layer_norm.bias"] = (TXT_WIDTH,)
    s["text_projection.weight"] = (PROJ_DIM, TXT_WIDTH)
    return s


def _draw(seed: int, name: str, shape: Tuple[int, ...]) -> np.ndarray:
    """One tensor. Scales are chosen so activations stay O(1) and softmax is not flat
    (a parity test on near-uniform attention would not exercise the softmax)."""
    g = _rng(seed, name)
    z = g.standard_normal(shape, dtype=np.float32)
    leaf = name.rsplit(".", 1)[-1]
    if len(shape) == 1:
        if "norm" in name and leaf == "weight":
            return (1.0 + 0.1 * z).astype(np.float32)
        if "norm" in name and leaf == "bias":
            return (0.05 * z).astype(np.float32)
        if name in ("outfit_token", "target_item_image_emb"):
            return (0.02 * z).astype(np.float32)
        if name.endswith("class_embedding"):
            return (0.5 * z).astype(np.float32)
        return (0.02 * z).astype(np.float32)  # biases
    if name.endswith("token_embedding.weight"):
        return (0.5 * z).astype(np.float32)
    if name.endswith("position_embedding.weight"):
        return (0.1 * z).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    gain = 0.5 if any(k in name for k in ("out_proj", "fc2", "linear2")) else 1.0
    return (z * np.float32(gain / np.sqrt(fan_in))).astype(np.float32)


def make_weights(seed: int, shapes: Dict[str, Tuple[int, ...]], prefix: str = "") -> Dict[str, np.ndarray]:
    """name → fp32 ndarray. `prefix` is prepended to the returned keys only; the random
    stream is keyed by the un-prefixed name so the same tensor is drawn either way."""
    return {prefix + k: _draw(seed, k, shp) for k, shp in shapes.items()}


def outfit_transformer_weights(seed: int) -> Dict[str, np.ndarray]:
    return make_weights(seed, outfit_transformer_shapes())


def vision_weights(seed: int, prefix: str = "", n_layers: int = VIT_LAYERS) -> Dict[str, np.ndarray]:
    return make_weights(seed, vision_shapes(n_layers), prefix)


def text_weights(seed: int, prefix: str = "", n_layers: int = TXT_LAYERS) -> Dict[str, np.ndarray]:
    return make_weights(seed, text_shapes(n_layers), prefix)


IMG_PREFIX = "item_encoder.image_enc.model."
TXT_PREFIX = "item_encoder.text_enc.model."


def full_state_dict(seed: int) -> Dict[str, np.ndarray]:
    """All 474 tensors / 202,432,625 parameters under the reference's key names."""
    sd = outfit_transformer_weights(seed)
    sd.update(vision_weights(seed, IMG_PREFIX))
    sd.update(text_weights(seed, TXT_PREFIX))
    return sd


def outlier_channels(sd: Dict[str, np.ndarray], level: int = 1) -> Dict[str, np.ndarray]:
    """A copy of a full state dict whose ViT carries massive activations, as trained CLIP ViTs do (a handful of hidden dimensions one
    to two orders of magnitude above the rest; every parity fixture otherwise comes from O(1) random-init weights):
      level 1: pre-LayerNorm gains and class-embedding entries of 3 residual-stream channels x 30, one fc2 output row (layer 3) x 20;
      level 2: those 3 channels x 100 (residual-stream values in the hundreds: with LayerNorm folding the RAW stream is the A operand
               of qkv and fc1), one fc2 output row x 50, and fc1 rows 77 of layers 2 and 6 (weight and bias) x 50, so that single
               hidden units of the MLP - fc2's A operand - run into the hundreds as well;
      level 3: level 2, and every encoder layer's LayerNorm gains AND the post-LayerNorm gain of the three massive channels x 1/100 - what trained
               networks do (without the post-LayerNorm part the pooled CLS row is dominated by the three channels, every image maps to nearly the same
               embedding and the logits stop depending on the ViT's arithmetic: levels 1 and 2 read 1-2e-5 for that reason): the raw stream
               carries the massive values (they dominate every LayerNorm's mean and variance, so the other channels arrive at a sixth of their usual
               scale), while their normalised contribution to the next GEMM is of ordinary size.  With LayerNorm folding the folded weight columns of
               those channels are 100x smaller than their neighbours and multiply operand values 100x larger."""
    sd = dict(sd)
    p = IMG_PREFIX + "vision_model."
    f_chan, f_row = (30.0, 20.0) if level == 1 else (100.0, 50.0)
    if level >= 3:
        for l in range(VIT_LAYERS):
            for ln in ("layer_norm1", "layer_norm2"):
                k = p + f"encoder.layers.{l}.{ln}.weight"
                v = sd[k].copy(); v[[5, 100, 700]] *= np.float32(0.01); sd[k] = v
        k = p + "post_layernorm.weight"
        v = sd[k].copy(); v[[5, 100, 700]] *= np.float32(0.01); sd[k] = v
    for k in (p + "pre_layrnorm.weight", p + "embeddings.class_embedding"):
        v = sd[k].copy(); v[[5, 100, 700]] *= np.float32(f_chan); sd[k] = v
    k = p + "encoder.layers.3.mlp.fc2.weight"
    v = sd[k].copy(); v[333] *= np.float32(f_row); sd[k] = v
    if level >= 2:
        for l in (2, 6):
            for leaf in ("weight", "bias"):
                k = p + f"encoder.layers.{l}.mlp.fc1.{leaf}"
                v = sd[k].copy(); v[77] *= np.float32(50.0); sd[k] = v
    return sd

def heavy_tailed(sd: Dict[str, np.ndarray], seed: int, df: float = 3.0) -> Dict[str, np.ndarray]:
    """A copy of a state dict whose matrices (every parameter with two or more dimensions) follow a Student-t distribution with `df` degrees of freedom
    at the variance the normal draw had: each element times sqrt(df / chi2_df) / sqrt(df / (df - 2)), the chi-square draws seeded by (seed, name).  At df = 3
    one weight in a thousand sits beyond 7 sigma and single weights of a row reach 20-60 sigma - the within-row dynamic range trained checkpoints have and
    O(1) normal draws do not: what the (hi, lo) split, the per-row scale of the fp8 lo copy and the LayerNorm-fold column sums meet."""
    assert df > 2
    out = dict(sd)
    for k, v in sd.items():
        if v.ndim >= 2:
            g = _rng(seed, "t:" + k)
            out[k] = (v * np.sqrt(df / g.chisquare(df, size=v.shape)) / np.sqrt(df / (df - 2.0))).astype(np.float32)
    return out

def sharp_attention(sd: Dict[str, np.ndarray], factor: float = 3.0) -> Dict[str, np.ndarray]:
    """A copy of a state dict whose attention logits are factor^2 times larger everywhere: q and k projection weights and biases of both CLIP towers and the
    q / k thirds of the set transformer's in_proj times `factor`.  Random-init attention is near-uniform (logits of order 0.1-1); trained attention is peaked,
    and a peaked softmax turns the operand rounding of q and k (2^-11 relative of the logit) into larger probability errors."""
    out = dict(sd)
    f = np.float32(factor)
    for k, v in sd.items():
        if k.endswith(("q_proj.weight", "q_proj.bias", "k_proj.weight", "k_proj.bias")):
            out[k] = (v * f).astype(np.float32)
        elif k.endswith(("in_proj_weight", "in_proj_bias")):
            w = v.copy(); n = w.shape[0] // 3
            w[:2 * n] *= f
            out[k] = w
    return out


def variant_state_dict(key) -> Dict[str, np.ndarray]:
    """Weights by fixture key: "<seed>" plain, "<seed>o<level>" with massive ViT channels (outlier_channels), "<seed>t<df>" heavy-tailed (heavy_tailed),
    "<seed>s<factor>" peaked attention (sharp_attention)."""
    key = str(key)
    if "s" in key:
        base, _, fac = key.partition("s")
        return sharp_attention(full_state_dict(int(base)), float(fac))
    if "o" in key:
        base, _, lvl = key.partition("o")
        return outlier_channels(full_state_dict(int(base)), int(lvl))
    if "t" in key:
        base, _, df = key.partition("t")
        return heavy_tailed(full_state_dict(int(base)), int(base), float(df))
    return full_state_dict(int(key))



# ---- inputs ---------------------------------------------------------------
def item_embeddings(seed: int, name: str, *lead: int) -> np.ndarray:
    """[*lead, 1024] fp32 rows shaped like ItemEncoder output: each 512-half L2-normalised
    (reference: base_image_encoder.py:46-47, base_text_encoder.py:37-38, model_utils.py:40-41)."""
    g = _rng(seed, name)
    z = g.standard_normal((*lead, 2, D_HALF), dtype=np.float32)
    z /= np.maximum(np.linalg.norm(z, axis=-1, keepdims=True), 1e-12)
    return z.reshape(*lead, D_MODEL).astype(np.float32)


def unit_rows(seed: int, name: str, *shape: int) -> np.ndarray:
    g = _rng(seed, name)
    z = g.standard_normal(shape, dtype=np.float32)
    z /= np.maximum(np.linalg.norm(z, axis=-1, keepdims=True), 1e-12)
    return z.astype(np.float32)


def outfit_batch(seed: int, B: int, L: int = 16, n_items=8, name: str = "outfits"):
    """(outfit_embedding [B,L,1024] fp32, outfit_mask [B,L] bool) exactly as the reference's
    collate emits them (outfit_x_base_processor.py:20-43): real rows first, zero pad rows after,
    mask True on the pad rows. `n_items` is an int or a length-B sequence."""
    n = np.full(B, n_items, dtype=np.int64) if np.isscalar(n_items) else np.asarray(n_items, dtype=np.int64)
    assert n.shape == (B,) and n.min() >= 0 and n.max() <= L
    emb = item_embeddings(seed, name, B, L)
    mask = np.arange(L)[None, :] >= n[:, None]
    emb[mask] = 0.0
    return emb, mask


def ragged_lengths(seed: int, B: int, lo: int, hi: int, name: str = "lengths") -> np.ndarray:
    return _rng(seed, name).integers(lo, hi + 1, size=B).astype(np.int64)


def pixel_values(seed: int, N: int, name: str = "pixels") -> np.ndarray:
    """[N,3,224,224] fp32: uniform uint8 images after the CLIP rescale+normalise that the
    reference's host-side CLIPImageProcessor applies (clip_image_encoder.py:29-31,69-71)."""
    g = _rng(seed, name)
    u8 = g.integers(0, 256, size=(N, 3, VIT_IMG, VIT_IMG), dtype=np.uint8)
    mean = np.asarray(CLIP_MEAN, np.float32).reshape(1, 3, 1, 1)
    std = np.asarray(CLIP_STD, np.float32).reshape(1, 3, 1, 1)
    return ((u8.astype(np.float32) * np.float32(1 / 255.0) - mean) / std).astype(np.float32)

def smooth_pixel_values(seed: int, N: int, name: str = "smooth") -> np.ndarray:
    """[N,3,224,224] fp32 images with the statistics photographs have and uniform noise does not: a per-image background colour, a coarse 7x7 random field
    upsampled bilinearly (large smooth regions), a few flat rectangles ("garments") and a little sensor noise, quantised to uint8 and put through the same
    CLIP rescale + normalise as pixel_values.  Neighbouring pixels are strongly correlated, so a 32x32 patch is nearly constant: the patch embedding sees
    inputs dominated by their common-mode component - the other end of the input distribution from uniform noise."""
    g = _rng(seed, name)
    out = np.empty((N, 3, VIT_IMG, VIT_IMG), dtype=np.float32)
    xs = (np.arange(VIT_IMG) + 0.5) / VIT_IMG * 6.0                     # sample positions in the 7-point coarse grid
    i0 = np.minimum(xs.astype(np.int64), 5); f = (xs - i0).astype(np.float32)
    for n in range(N):
        bg = g.uniform(40, 230, size=(3, 1, 1)).astype(np.float32)
        coarse = g.normal(0, 35, size=(3, 7, 7)).astype(np.float32)
        rows = coarse[:, i0, :] * (1 - f)[None, :, None] + coarse[:, i0 + 1, :] * f[None, :, None]          # [3, 224, 7]
        img = bg + rows[:, :, i0] * (1 - f)[None, None, :] + rows[:, :, i0 + 1] * f[None, None, :]
        for _ in range(int(g.integers(1, 4))):
            y0, x0 = (int(v) for v in g.integers(0, 160, size=2)); h, w = (int(v) for v in g.integers(30, 120, size=2))
            img[:, y0:y0 + h, x0:x0 + w] = g.uniform(0, 255, size=(3, 1, 1)).astype(np.float32)
        img += g.normal(0, 2.0, size=img.shape).astype(np.float32)
        out[n] = np.clip(np.rint(img), 0, 255)
    mean = np.asarray(CLIP_MEAN, np.float32).reshape(1, 3, 1, 1)
    std = np.asarray(CLIP_STD, np.float32).reshape(1, 3, 1, 1)
    return ((out * np.float32(1 / 255.0) - mean) / std).astype(np.float32)


def token_batch(seed: int, N: int, T: int = 64, n_real=8, name: str = "tokens"):
    """(input_ids [N,T] int64, attention_mask [N,T] int64) shaped like
    CLIPTokenizer(max_length=64, padding='max_length') output (clip_text_encoder.py:42-50):
    BOS, words, EOS, then EOS-id padding with attention_mask 0."""
    g = _rng(seed, name)
    n = np.full(N, n_real, dtype=np.int64) if np.isscalar(n_real) else np.asarray(n_real, dtype=np.int64)
    assert n.min() >= 2 and n.max() <= T
    ids = np.full((N, T), EOS_ID, dtype=np.int64)
    ids[:, 0] = BOS_ID
    words = g.integers(1000, 40000, size=(N, T), dtype=np.int64)
    pos = np.arange(T)[None, :]
    inner = (pos >= 1) & (pos < (n[:, None] - 1))
    ids[inner] = words[inner]
    att = (pos < n[:, None]).astype(np.int64)
    return ids, att


def bench_batch(seed: int, B: int = 256, n_items: int = 8, T: int = 64, n_real: int = 8):
    """bench.py's batch (BASELINE configs[1]), generated on the HOST so that the build container (where the reference can be
    imported: oracle/gen_bench_golden.py) and the GPU box see the same inputs: pixel_values [B, n, 3, 224, 224] fp32 (uniform
    uint8 after CLIP rescale + normalise), input_ids / attention_mask [B * n, T] (BOS + words + EOS, padded)."""
    px = pixel_values(seed, B * n_items).reshape(B, n_items, 3, VIT_IMG, VIT_IMG)
    ids, att = token_batch(seed, B * n_items, T, n_real)
    return px, ids, att

# ---- a synthetic CLIP BPE vocabulary (string inputs) -------------------------------------------------
# The fashion-clip vocabulary is not available offline.  CLIPTokenizer itself is (transformers): given ANY vocab.json + merges.txt of
# CLIP's format it runs the reference's string path (clip_text_encoder.py:42-50: lower-case, byte-level BPE, BOS / EOS, EOS-id padding
# to 64).  This writes such a pair, deterministically: the 256 byte symbols and their end-of-word forms (ids 0..511, as in the real
# vocabulary), one merged token per rule below, fillers up to 49,405, BOS 49,406, EOS 49,407 (the ids the text tower's kernels key on).
CLIP_SYNTH_MERGES = [("r", "e"), ("re", "d</w>"), ("d", "r"), ("dr", "e"), ("dre", "s"), ("dres", "s</w>"), ("b", "l"), ("bl", "u"), ("blu", "e</w>"),
                     ("j", "e"), ("je", "a"), ("jea", "n"), ("jean", "s</w>"), ("s", "h"), ("sh", "o"), ("sho", "e"), ("shoe", "s</w>"),
                     ("t", "o"), ("to", "p</w>"), ("s", "k"), ("sk", "i"), ("ski", "r"), ("skir", "t</w>"), ("b", "a"), ("ba", "g</w>"),
                     ("c", "o"), ("co", "a"), ("coa", "t</w>"), ("h", "a"), ("ha", "t</w>"), ("l", "e"), ("le", "a"), ("lea", "t"),
                     ("leat", "h"), ("leath", "e"), ("leathe", "r</w>"), ("w", "o"), ("wo", "o"), ("woo", "l</w>")]


def _bytes_to_unicode() -> Dict[int, str]:
    """The byte -> printable-character table of GPT-2 / CLIP byte-level BPE (public algorithm; transformers 5 no longer exports it)."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b); cs.append(256 + n); n += 1
    return dict(zip(bs, (chr(c) for c in cs)))


def write_clip_vocabulary(directory: str) -> str:
    """vocab.json + merges.txt in `directory` (created); returns it.  CLIPTokenizer.from_pretrained(directory) then tokenises strings."""
    import json, os
    chars = list(_bytes_to_unicode().values())
    vocab: Dict[str, int] = {}
    for c in chars:
        vocab[c] = len(vocab)
    for c in chars:
        vocab[c + "</w>"] = len(vocab)
    for a, b in CLIP_SYNTH_MERGES:
        vocab.setdefault(a + b, len(vocab))
    while len(vocab) < BOS_ID:
        vocab[f"<filler_{len(vocab)}>"] = len(vocab)
    vocab["<|startoftext|>"] = BOS_ID
    vocab["<|endoftext|>"] = EOS_ID
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, "vocab.json"), "w") as f:
        json.dump(vocab, f)
    with open(os.path.join(directory, "merges.txt"), "w") as f:
        f.write("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in CLIP_SYNTH_MERGES) + "\n")
    return directory


def checksum(a: np.ndarray) -> str:
    return f"{zlib.crc32(np.ascontiguousarray(a).tobytes()):08x}"
