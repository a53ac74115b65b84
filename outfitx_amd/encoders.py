"""Item encoder of the scoring path: CLIP ViT-B/32 image tower + CLIP text tower + fuser, with the
reference's module tree (so `state_dict()` keys and strict `load_state_dict` match SURVEY.md §8b)
but with every forward running in libofx_hip.so.

Reference: src/models/encoders/item_encoder.py:8-61, base_encoders/base_image_encoder.py:17-49,
base_encoders/base_text_encoder.py:14-40, image_encoders/clip_image_encoder.py:10-79,
text_encoders/clip_text_encoder.py:11-60, src/utils/model_utils.py:26-48.

The nn.Linear / nn.LayerNorm / nn.Embedding / nn.Conv2d objects below are PARAMETER CONTAINERS
(names, shapes, default init); their torch forward is never called.
"""
from __future__ import annotations

import ctypes as C
import itertools
import warnings
from types import SimpleNamespace
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
from PIL import Image
from torch import nn

from . import _lib as L
from .configs import ItemEncoderConfig
from .engine import Engine

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def freeze_model(model: nn.Module) -> None:
    """src/utils/model_utils.py:8-10"""
    for p in model.parameters():
        p.requires_grad = False


def flatten_seq_to_one_dim(sequences: List[List[Any]]) -> List[Any]:
    """src/utils/model_utils.py:47-48"""
    return list(itertools.chain(*sequences))


def aggregate_embeddings(image_embeddings=None, text_embeddings=None, aggregation_method: str = "concat"):
    """src/utils/model_utils.py:26-45.  'mean' keeps the reference's literal tensor semantics
    (mean over dim=-2 of the stacked pair -> [2,B,512]); see SURVEY.md §7.3."""
    embeds = [e for e in (image_embeddings, text_embeddings) if e is not None]
    if not embeds:
        raise ValueError("At least one of image_embeds or text_embeds must be provided.")
    if aggregation_method == "concat":
        return torch.cat(embeds, dim=-1)
    if aggregation_method == "mean":
        return torch.mean(torch.stack(embeds), dim=-2)
    raise ValueError(f"Unsupported aggregation method: {aggregation_method}. Use 'concat' or 'mean'.")


# ----------------------------------------------------------------------------- parameter trees
class _ClipAttentionParams(nn.Module):
    def __init__(self, width):
        super().__init__()
        self.k_proj = nn.Linear(width, width)
        self.v_proj = nn.Linear(width, width)
        self.q_proj = nn.Linear(width, width)
        self.out_proj = nn.Linear(width, width)


class _ClipMlpParams(nn.Module):
    def __init__(self, width, mlp):
        super().__init__()
        self.fc1 = nn.Linear(width, mlp)
        self.fc2 = nn.Linear(mlp, width)


class _ClipLayerParams(nn.Module):
    def __init__(self, width, mlp, eps):
        super().__init__()
        self.self_attn = _ClipAttentionParams(width)
        self.layer_norm1 = nn.LayerNorm(width, eps=eps)
        self.mlp = _ClipMlpParams(width, mlp)
        self.layer_norm2 = nn.LayerNorm(width, eps=eps)


class _ClipEncoderParams(nn.Module):
    def __init__(self, width, mlp, n_layers, eps):
        super().__init__()
        self.layers = nn.ModuleList([_ClipLayerParams(width, mlp, eps) for _ in range(n_layers)])


def _layer_tensors(layer: _ClipLayerParams) -> List[torch.Tensor]:
    a, m = layer.self_attn, layer.mlp
    return [a.k_proj.weight, a.k_proj.bias, a.v_proj.weight, a.v_proj.bias, a.q_proj.weight, a.q_proj.bias,
            a.out_proj.weight, a.out_proj.bias, layer.layer_norm1.weight, layer.layer_norm1.bias,
            m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, layer.layer_norm2.weight, layer.layer_norm2.bias]


class _VisionEmbeddingParams(nn.Module):
    def __init__(self, width, patch, image):
        super().__init__()
        self.class_embedding = nn.Parameter(torch.randn(width))
        self.patch_embedding = nn.Conv2d(3, width, kernel_size=patch, stride=patch, bias=False)
        self.position_embedding = nn.Embedding((image // patch) ** 2 + 1, width)


class _VisionTransformerParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.embeddings = _VisionEmbeddingParams(c.hidden_size, c.patch_size, c.image_size)
        self.pre_layrnorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)   # sic: upstream HF spelling
        self.encoder = _ClipEncoderParams(c.hidden_size, c.intermediate_size, c.num_hidden_layers, c.layer_norm_eps)
        self.post_layernorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class ClipVisionParams(nn.Module):
    """Same parameter names as HF CLIPVisionModelWithProjection (ViT-B/32 defaults)."""

    def __init__(self, **over):
        super().__init__()
        self.config = SimpleNamespace(hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                                      num_attention_heads=12, patch_size=32, image_size=224, projection_dim=512,
                                      hidden_act="quick_gelu", layer_norm_eps=1e-5)
        self.config.__dict__.update(over)
        self.vision_model = _VisionTransformerParams(self.config)
        self.visual_projection = nn.Linear(self.config.hidden_size, self.config.projection_dim, bias=False)

    def pack_list(self) -> List[torch.Tensor]:
        v = self.vision_model
        out = [v.embeddings.class_embedding, v.embeddings.patch_embedding.weight, v.embeddings.position_embedding.weight,
               v.pre_layrnorm.weight, v.pre_layrnorm.bias]
        for l in v.encoder.layers:
            out += _layer_tensors(l)
        return out + [v.post_layernorm.weight, v.post_layernorm.bias, self.visual_projection.weight]


class _TextEmbeddingParams(nn.Module):
    def __init__(self, vocab, max_pos, width):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab, width)
        self.position_embedding = nn.Embedding(max_pos, width)


class _TextTransformerParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.embeddings = _TextEmbeddingParams(c.vocab_size, c.max_position_embeddings, c.hidden_size)
        self.encoder = _ClipEncoderParams(c.hidden_size, c.intermediate_size, c.num_hidden_layers, c.layer_norm_eps)
        self.final_layer_norm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class ClipTextParams(nn.Module):
    """Same parameter names as HF CLIPTextModelWithProjection (CLIP B/32 text defaults)."""

    def __init__(self, **over):
        super().__init__()
        self.config = SimpleNamespace(hidden_size=512, intermediate_size=2048, num_hidden_layers=12,
                                      num_attention_heads=8, vocab_size=49408, max_position_embeddings=77,
                                      projection_dim=512, hidden_act="quick_gelu", layer_norm_eps=1e-5,
                                      eos_token_id=49407)
        self.config.__dict__.update(over)
        self.text_model = _TextTransformerParams(self.config)
        self.text_projection = nn.Linear(self.config.hidden_size, self.config.projection_dim, bias=False)

    def pack_list(self) -> List[torch.Tensor]:
        t = self.text_model
        out = [t.embeddings.token_embedding.weight, t.embeddings.position_embedding.weight]
        for l in t.encoder.layers:
            out += _layer_tensors(l)
        return out + [t.final_layer_norm.weight, t.final_layer_norm.bias, self.text_projection.weight]


def _try_load_pretrained(params: nn.Module, hf_class_name: str, name: str) -> bool:
    """Copy a HF checkpoint's tensors into our parameter tree when one is available offline/online.
    HF is used as a FILE READER only; it never computes anything here."""
    try:
        import transformers
        hf = getattr(transformers, hf_class_name).from_pretrained(name, local_files_only=True)
        for k in ("hidden_act", "layer_norm_eps", "eos_token_id"):
            if hasattr(hf.config, k) and hasattr(params.config, k):
                setattr(params.config, k, getattr(hf.config, k))
        params.load_state_dict(hf.state_dict(), strict=True)
        return True
    except Exception:
        return False


# ----------------------------------------------------------------------------- host preprocessing
def clip_preprocess(images: Sequence[Union[np.ndarray, Image.Image]], size: int = 224) -> torch.Tensor:
    """Host-side equivalent of CLIPImageProcessor(do_convert_rgb=False) as the reference uses it
    (clip_image_encoder.py:29-31,69-71): resize shortest edge to `size` (bicubic), centre-crop,
    rescale 1/255, normalise with the CLIP mean/std.  -> [N,3,size,size] fp32."""
    out = np.empty((len(images), 3, size, size), np.float32)
    mean = np.asarray(CLIP_MEAN, np.float32).reshape(3, 1, 1)
    std = np.asarray(CLIP_STD, np.float32).reshape(3, 1, 1)
    for i, im in enumerate(images):
        if not isinstance(im, Image.Image):
            im = Image.fromarray(np.asarray(im))
        w, h = im.size
        short, long_ = (w, h) if w <= h else (h, w)
        new_short, new_long = size, int(size * long_ / short)
        nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
        if (nw, nh) != (w, h):
            im = im.resize((nw, nh), resample=Image.BICUBIC)
        left, top = (nw - size) // 2, (nh - size) // 2
        im = im.crop((left, top, left + size, top + size))
        a = np.asarray(im, np.float32)
        if a.ndim == 2:
            a = np.repeat(a[:, :, None], 3, 2)
        out[i] = (a.transpose(2, 0, 1) * np.float32(1 / 255.0) - mean) / std
    return torch.from_numpy(out)


# ----------------------------------------------------------------------------- encoders
class _TowerBase(nn.Module):
    """Engine plumbing shared by the two towers: an Engine per device, repacked when parameters change."""

    def __init__(self):
        super().__init__()
        self._engines: Dict[Any, Engine] = {}
        self.tower_precision = L.DEFAULT_TOWER_PRECISION

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    def _signature(self):
        return tuple((p.data_ptr(), p._version) for p in self.model.parameters())

    def _engine(self, kind: str) -> Engine:
        dev = self.device
        key = (dev, self.tower_precision)
        eng = self._engines.get(key)
        if eng is None:
            eng = Engine(dev, self._desc(), tower_precision=self.tower_precision)
            self._engines[key] = eng
        sig = self._signature()
        if eng.signature[kind] != sig:
            getattr(eng, "pack_" + kind)(self.model.pack_list())
            eng.signature[kind] = sig
        return eng

    def __getstate__(self):          # engines hold ctypes handles: never pickle / deepcopy them
        s = self.__dict__.copy()
        s["_engines"] = {}
        return s


class CLIPImageEncoder(_TowerBase):
    """Reference: image_encoders/clip_image_encoder.py:10-79 (+ BaseImageEncoder.forward)."""

    def __init__(self, model_name_or_path: str = "patrickjohncyh/fashion-clip", freeze: bool = True):
        super().__init__()
        self.model = ClipVisionParams()
        self.pretrained = _try_load_pretrained(self.model, "CLIPVisionModelWithProjection", model_name_or_path)
        if not self.pretrained:
            warnings.warn(f"CLIP vision checkpoint '{model_name_or_path}' not available offline: random-initialised "
                          "(load weights with load_state_dict)", stacklevel=2)
        self.model.eval()
        if freeze:
            freeze_model(self.model)
        self.processor = SimpleNamespace(size={"shortest_edge": self.model.config.image_size})

    def _desc(self) -> L.ModelDesc:
        d, c = L.default_desc(), self.model.config
        d.vit_width, d.vit_layers, d.vit_heads, d.vit_mlp = c.hidden_size, c.num_hidden_layers, c.num_attention_heads, c.intermediate_size
        d.vit_patch, d.vit_image, d.proj_dim, d.ln_eps = c.patch_size, c.image_size, c.projection_dim, c.layer_norm_eps
        d.vit_act = L.ACTS[c.hidden_act]
        return d

    @property
    def image_size(self) -> Tuple[int, int]:
        s = self.processor.size["shortest_edge"]
        return (s, s)

    @property
    def d_embed(self) -> int:
        return self.model.config.projection_dim

    def _pixels(self, images):
        """-> (pixel_values tensor or None, list of uint8 arrays or None, B).  uint8 / PIL input on a HIP device is handed
        to the GPU preprocessor as packed bytes; anything else goes through the host pipeline."""
        if isinstance(images, torch.Tensor):
            b = images.size(0)
            return images.reshape(b * images.size(1), *images.shape[2:]), None, b
        if len(set(len(seq) for seq in images)) != 1:
            raise ValueError("All sequences in images should have the same length.")
        flat = flatten_seq_to_one_dim(images)
        arrs = [np.asarray(im if not isinstance(im, Image.Image) or im.mode in ("RGB", "L") else im.convert("RGB")) for im in flat]
        if all(a.dtype == np.uint8 for a in arrs) and self.device.type == "cuda":
            return None, arrs, len(images)
        return clip_preprocess(flat, self.model.config.image_size), None, len(images)

    # > 1: a tensor batch of >= 512 images is split into that many contiguous parts, each run on its own HIP stream with its own
    # workspace.  The tower's GEMMs are one 160 KB-LDS block per CU, so a kernel's last, partly filled round of tiles (6 % of a
    # 1,200-tile out-proj / fc2 launch on 256 CUs) and its store-burst epilogues leave CUs idle that the other stream's kernels fill.
    vit_streams = 1

    def _run(self, px, arrs, out: torch.Tensor, col: int, normalize: bool) -> None:
        eng = self._engine("vision")
        if arrs is None and self.vit_streams > 1 and px.shape[0] >= 512:
            cur = torch.cuda.current_stream(self.device)
            pool = getattr(self, "_side_streams", None)
            if pool is None or len(pool) < self.vit_streams:
                pool = self._side_streams = [torch.cuda.Stream(self.device) for _ in range(self.vit_streams)]
            n = px.shape[0]
            step = -(-n // self.vit_streams)
            for i in range(self.vit_streams):
                a, b = i * step, min(n, (i + 1) * step)
                if a >= b:
                    continue
                pool[i].wait_stream(cur)
                with torch.cuda.stream(pool[i]):
                    eng.vit(px[a:b], out[a:b], col, normalize)
            for i in range(self.vit_streams):
                cur.wait_stream(pool[i])
        elif arrs is None:
            eng.vit(px, out, col, normalize)
        elif self.fused_preprocess:      # resample -> patch-embedding operand, no pixel tensor (ofx_vit_b32_fwd_u8)
            eng.vit_u8(arrs, CLIP_MEAN, CLIP_STD, out, col, normalize)
        else:
            eng.vit(eng.clip_preprocess(arrs, self.model.config.image_size, CLIP_MEAN, CLIP_STD), out, col, normalize)

    fused_preprocess = True

    @torch.no_grad()
    def encode_into(self, images, out: torch.Tensor, col: int, normalize: bool) -> int:
        """Run the tower and write [N,512] into out[:, col:col+512]; returns the batch size B."""
        px, arrs, b = self._pixels(images)
        self._run(px, arrs, out, col, normalize)
        return b

    @torch.no_grad()
    def forward(self, images, normalize: bool = True, *args, **kwargs) -> torch.Tensor:
        px, arrs, b = self._pixels(images)
        n = px.shape[0] if arrs is None else len(arrs)
        out = torch.empty(n, self.d_embed, dtype=torch.float32, device=self.device)
        self._run(px, arrs, out, 0, normalize)
        return out.view(b, -1, self.d_embed)


class CLIPTextEncoder(_TowerBase):
    """Reference: text_encoders/clip_text_encoder.py:11-60 (+ BaseTextEncoder.forward).  Accepts
    List[List[str]] (needs the CLIP BPE vocabulary to be available to transformers) or an already
    tokenised dict {'input_ids','attention_mask'} of [B,L,T] tensors."""

    def __init__(self, model_name_or_path: str = "patrickjohncyh/fashion-clip", freeze: bool = True):
        super().__init__()
        self.model = ClipTextParams()
        self.pretrained = _try_load_pretrained(self.model, "CLIPTextModelWithProjection", model_name_or_path)
        if not self.pretrained:
            warnings.warn(f"CLIP text checkpoint '{model_name_or_path}' not available offline: random-initialised",
                          stacklevel=2)
        self.model.eval()
        if freeze:
            freeze_model(self.model)
        self._tok_name = model_name_or_path
        self.tokenizer = None
        self.dedup_texts = False      # opt-in: run the tower once per DISTINCT token row of a call (item texts are category names)
        # opt-in, SURVEY.md section 8f N2: a PERSISTENT token-row -> embedding cache (Polyvore's item texts are 132 category strings): the
        # tower runs on rows no earlier call has seen, everything else is a row gather from a device-resident table.  Dropped whenever a
        # parameter of the tower changes (tensor version counters) or the normalize flag differs; host-resident token ids only.
        self.cache_texts = False
        self._cache_rows: dict = {}
        self._cache_table: Optional[torch.Tensor] = None
        self._cache_state = None
        self.cache_tower_rows = 0     # rows the tower has computed through the cache (tests / diagnostics)

    def _desc(self) -> L.ModelDesc:
        d, c = L.default_desc(), self.model.config
        d.txt_width, d.txt_layers, d.txt_heads, d.txt_mlp = c.hidden_size, c.num_hidden_layers, c.num_attention_heads, c.intermediate_size
        d.txt_vocab, d.txt_max_pos, d.proj_dim, d.ln_eps = c.vocab_size, c.max_position_embeddings, c.projection_dim, c.layer_norm_eps
        d.txt_act, d.txt_eos_id = L.ACTS[c.hidden_act], c.eos_token_id
        return d

    @property
    def d_embed(self) -> int:
        return self.model.config.projection_dim

    def _tokenize(self, texts: List[str], tokenizer_kargs: Optional[dict]):
        if self.tokenizer is None:
            try:
                from transformers import CLIPTokenizer
                self.tokenizer = CLIPTokenizer.from_pretrained(self._tok_name)
            except Exception as e:
                raise RuntimeError("CLIP tokenizer files are not available; pass pre-tokenised "
                                   "{'input_ids','attention_mask'} tensors instead of strings") from e
        kw = tokenizer_kargs if tokenizer_kargs is not None else {"max_length": 64, "padding": "max_length", "truncation": True}
        kw["return_tensors"] = "pt"
        enc = self.tokenizer(text=texts, **kw)
        return enc["input_ids"], enc["attention_mask"]

    def _ids(self, texts, tokenizer_kargs=None):
        if isinstance(texts, dict):
            ids = texts["input_ids"]
            b, l = ids.size(0), ids.size(1)
            att = texts.get("attention_mask")
            return ids.reshape(b * l, -1), None if att is None else att.reshape(b * l, -1), b
        if len(set(len(seq) for seq in texts)) != 1:
            raise ValueError("All sequences in texts should have the same length.")
        ids, att = self._tokenize(flatten_seq_to_one_dim(texts), tokenizer_kargs)
        return ids, att, len(texts)

    def _lengths(self, ids: torch.Tensor):
        """EOS position + 1 per text when the ids live on the host (no device sync otherwise)."""
        if ids.device.type != "cpu":
            return None
        eos = self.model.config.eos_token_id
        pos = ids.argmax(-1) if eos == 2 else (ids == eos).int().argmax(-1)
        return (pos + 1).tolist()

    def _dedup(self, ids: torch.Tensor, att: Optional[torch.Tensor]):
        """dedup_texts (the item texts are 132 category strings): host-side unique over the token rows ->
        (unique ids, unique mask, inverse index) or None when nothing repeats / the ids are not on the host."""
        if not self.dedup_texts or ids.device.type != "cpu" or ids.shape[0] < 2:
            return None
        a = ids.numpy()
        key = a if att is None else a * 2 + att.numpy().astype(a.dtype)          # ids < 2^31: (id, mask bit) pairs stay distinct
        w = getattr(self, "_hash_w", None)
        if w is None or w.shape[0] != key.shape[1]:
            w = np.random.default_rng(0x5EED).integers(1, 2 ** 62, key.shape[1], dtype=np.int64) | 1
            self._hash_w = w
        h = (key.astype(np.int64) * w).sum(1)                                      # wrap-around 64-bit row hash
        _, first, inv = np.unique(h, return_index=True, return_inverse=True)
        if first.shape[0] == key.shape[0] or not np.array_equal(key[first][inv], key):   # nothing repeats / a hash collision: plain path
            return None
        first_t = torch.from_numpy(first)
        uniq_ids = ids.index_select(0, first_t)
        uniq_att = None if att is None else att.index_select(0, first_t)
        return uniq_ids.contiguous(), (None if uniq_att is None else uniq_att.contiguous()), torch.from_numpy(inv.astype(np.int64))

    def _cache_plan(self, ids: torch.Tensor, att: Optional[torch.Tensor]):
        """cache_texts: (table row per input row [n] int64, indices of the first occurrence of every row the table lacks, their keys,
        first new slot).  Nothing is registered here: the new keys enter `_cache_rows` only once the tower has filled their slots
        (`_run`), so a plan that is never run - or whose tower call raises - leaves no key pointing at an uninitialised table row."""
        a = ids.numpy()
        key = a if att is None else a * 2 + att.numpy().astype(a.dtype)
        rows = np.empty(a.shape[0], np.int64)
        slot0 = len(self._cache_rows)
        miss, new_keys, pending = [], [], {}
        for i in range(a.shape[0]):
            k = key[i].tobytes()
            r = self._cache_rows.get(k)
            if r is None:
                r = pending.get(k)
                if r is None:
                    r = pending[k] = slot0 + len(miss)
                    miss.append(i)
                    new_keys.append(k)
            rows[i] = r
        return rows, miss, new_keys, slot0

    @torch.no_grad()
    def prepare(self, texts, tokenizer_kargs=None):
        """Host part of the text path (tokenise, optional de-duplication, EOS lengths, stage ids on the device).  ItemEncoder
        calls it before the image tower is enqueued so no blocking copy sits in the middle of the step."""
        ids, att, b = self._ids(texts, tokenizer_kargs)
        inverse = None
        if self.cache_texts and ids.device.type == "cpu":
            state = tuple(p._version for p in self.model.parameters())
            if state != self._cache_state:                                          # the tower's weights changed: every cached row is stale
                self._cache_rows, self._cache_table, self._cache_state = {}, None, state
            rows, miss, new_keys, slot0 = self._cache_plan(ids, att)
            plan = {"rows": torch.from_numpy(rows).to(self.device, non_blocking=True), "slot0": slot0, "n_new": len(miss), "new_keys": new_keys}
            if miss:
                sel = torch.as_tensor(miss, dtype=torch.long)
                m_ids = ids.index_select(0, sel).contiguous()
                m_att = None if att is None else att.index_select(0, sel).contiguous()
                plan["lengths"] = self._lengths(m_ids)
                plan["ids"], plan["att"] = self._engine("text").stage_tokens(m_ids, m_att)
            return None, None, None, b, plan
        dd = self._dedup(ids, att)
        if dd is not None:
            ids, att, inverse = dd
            inverse = inverse.to(self.device, non_blocking=True)
        lengths = self._lengths(ids)
        ids_d, att_d = self._engine("text").stage_tokens(ids, att)
        return ids_d, att_d, lengths, b, inverse

    def _run(self, ids, att, lengths, inverse, out: torch.Tensor, col: int, normalize: bool) -> None:
        eng = self._engine("text")
        if isinstance(inverse, dict):                   # persistent cache: the tower on the new rows only, into their table slots; then one gather
            plan, d = inverse, self.d_embed
            if self._cache_table is not None and getattr(self, "_cache_norm", normalize) != normalize:
                raise ValueError("cache_texts: the cache was filled with normalize=%s; clear it (set cache_texts again) before changing the flag" % (not normalize))
            self._cache_norm = normalize
            if plan["slot0"] != len(self._cache_rows):
                raise RuntimeError("cache_texts: this prepared batch is stale (another batch was prepared and run, or the cache was cleared, since prepare()); prepare it again")
            need = plan["slot0"] + plan["n_new"]
            if self._cache_table is None or self._cache_table.shape[0] < need:
                grown = torch.empty(max(need, 256, 2 * (0 if self._cache_table is None else self._cache_table.shape[0])), d, dtype=torch.float32, device=self.device)
                if self._cache_table is not None:
                    grown[:self._cache_table.shape[0]] = self._cache_table
                self._cache_table = grown
            if plan["n_new"]:
                eng.text(plan["ids"], plan["att"], self._cache_table[plan["slot0"]:need], 0, normalize, plan["lengths"])
                for j, k in enumerate(plan["new_keys"]):                  # registered only now: their table rows exist (in stream order)
                    self._cache_rows[k] = plan["slot0"] + j
                self.cache_tower_rows += plan["n_new"]
            out[:, col:col + d] = self._cache_table.index_select(0, plan["rows"])
            return
        if inverse is None:
            eng.text(ids, att, out, col, normalize, lengths)
            return
        uniq = torch.empty(ids.shape[0], self.d_embed, dtype=torch.float32, device=self.device)     # the tower runs on the distinct texts only
        eng.text(ids, att, uniq, 0, normalize, lengths)
        out[:, col:col + self.d_embed] = uniq.index_select(0, inverse)

    @torch.no_grad()
    def encode_into(self, texts, out: torch.Tensor, col: int, normalize: bool, tokenizer_kargs=None, prepared=None) -> int:
        ids, att, lengths, b, inverse = prepared if prepared is not None else self.prepare(texts, tokenizer_kargs)
        self._run(ids, att, lengths, inverse, out, col, normalize)
        return b

    @torch.no_grad()
    def forward(self, texts, normalize: bool = True, *args, **kwargs) -> torch.Tensor:
        ids, att, lengths, b, inverse = self.prepare(texts, kwargs.get("tokenizer_kargs"))
        n = inverse["rows"].shape[0] if isinstance(inverse, dict) else (ids.shape[0] if inverse is None else inverse.shape[0])
        out = torch.empty(n, self.d_embed, dtype=torch.float32, device=self.device)
        self._run(ids, att, lengths, inverse, out, 0, normalize)
        return out.view(b, -1, self.d_embed)


class ItemEncoder(nn.Module):
    """Reference: src/models/encoders/item_encoder.py:8-61 — only the type='clip' branch (:20-26) is built."""

    def __init__(self, cfg: ItemEncoderConfig):
        super().__init__()
        self.cfg = cfg
        if cfg.type != "clip":
            raise NotImplementedError(
                f"ItemEncoderConfig(type='{cfg.type}') is outside the MI355X scoring path; build the model with "
                "OutfitXConfig(item_encoder=ItemEncoderConfig(type='clip'))")
        self.image_enc = CLIPImageEncoder(model_name_or_path=cfg.clip_model_name)
        self.text_enc = CLIPTextEncoder(model_name_or_path=cfg.clip_model_name)
        # side-stream text tower: its small kernels fill the tile-quantisation tails of the big ViT GEMMs.  Neutral with round 1's
        # 1.3 ms single-product text tower; with the three-product one (5 ms) 31.85 vs 32.85 ms per cfg2 step (bench.py --overlap-towers)
        self.overlap_towers = True
        self.side_stream_priority = "normal"        # "low": a lowest-priority HIP stream for the text tower (tools/overlap_ab.py)
        self._streams: Dict[Any, Any] = {}

    def _side_stream(self, dev):
        key = (dev, self.side_stream_priority)
        s = self._streams.get(key)
        if s is None:
            if self.side_stream_priority == "low":
                # lowest dispatch priority (torch's own pools stop at HIP's normal level): the text tower's workgroups take CUs
                # only when the ViT's stream has none ready.  The raw stream lives as long as the process (one per device).
                raw = C.c_void_p()
                L.check(L.load().ofx_stream_create_low_priority(dev.index if dev.index is not None else torch.cuda.current_device(), C.byref(raw)),
                        "ofx_stream_create_low_priority")
                s = torch.cuda.ExternalStream(raw.value, device=dev)
            else:
                s = torch.cuda.Stream(device=dev)
            self._streams[key] = s
        return s

    def __getstate__(self):
        s = self.__dict__.copy()
        s["_streams"] = {}
        return s

    @property
    def d_embed(self) -> int:
        return self.cfg.dim_per_modality * 2 if self.cfg.aggregation_method == "concat" else self.cfg.dim_per_modality

    @property
    def image_size(self):
        return self.image_enc.image_size

    def set_precision(self, tower_precision: str) -> None:
        self.image_enc.tower_precision = tower_precision
        self.text_enc.tower_precision = tower_precision

    def forward(self, images, texts, *args, prepared_texts=None, **kwargs) -> torch.Tensor:
        """prepared_texts: the tuple CLIPTextEncoder.prepare returns, for callers that staged the tokens themselves (graphs.ForwardReplay)."""
        if self.cfg.aggregation_method == "concat":
            # concat fuser fused into the towers' epilogues: both write straight into one [B*L,1024] buffer
            dev = self.image_enc.device
            n = (images.size(0) * images.size(1)) if isinstance(images, torch.Tensor) else sum(len(s) for s in images)
            d = self.cfg.dim_per_modality
            out = torch.empty(n, 2 * d, dtype=torch.float32, device=dev)
            prepared = prepared_texts if prepared_texts is not None else self.text_enc.prepare(texts)   # host work + H2D first, then both towers are enqueued back to back
            if self.overlap_towers and dev.type == "cuda":
                # the text tower is independent of the image tower: on a side HIP stream its small kernels fill the
                # tile-quantisation tails of the big ViT GEMMs; both write disjoint columns of `out`
                main = torch.cuda.current_stream(dev)
                side = self._side_stream(dev)
                side.wait_stream(main)
                # No record_stream on `out` / the prepared token tensors: they are allocated on `main`, and `main` joins `side` below before
                # anything later on `main` can run - so when their blocks return to main's pool and are handed out again, that use is
                # ordered after the side stream's last access.  (record_stream would defer every free until the device has caught up with
                # the host, which runs steps ahead: the allocator then keeps calling hipMalloc inside steady-state steps - 7-9 calls per 10
                # steps in bench.py, +1-2 ms each and now and then a step at half speed.)
                try:
                    with torch.cuda.stream(side):
                        b2 = self.text_enc.encode_into(texts, out, d, self.cfg.norm_out, prepared=prepared)
                    b = self.image_enc.encode_into(images, out, 0, self.cfg.norm_out)
                finally:
                    main.wait_stream(side)                     # the join the comment above relies on, also when either tower raises
            else:
                b = self.image_enc.encode_into(images, out, 0, self.cfg.norm_out)
                b2 = self.text_enc.encode_into(texts, out, d, self.cfg.norm_out, prepared=prepared)
            if b != b2:
                raise ValueError("images and texts disagree on the batch size")
            return out.view(b, -1, 2 * d)
        img = self.image_enc(images, normalize=self.cfg.norm_out, *args, **kwargs)
        txt = self.text_enc(texts, normalize=self.cfg.norm_out, *args, **kwargs)
        return aggregate_embeddings(image_embeddings=img, text_embeddings=txt, aggregation_method=self.cfg.aggregation_method)

    encode_items = forward   # north-star alias (the reference has no encode_items)
