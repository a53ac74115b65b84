"""OutfitX — drop-in for the reference's `src.models.OutfitX` (src/models/outfit_x.py:17-172) whose
forward runs on hand-written gfx950 kernels through libofx_hip.so.

Same constructor, same `forward(task, **tensors)` dispatch on the task CLASS, same attribute
names and the same `state_dict()` key set / shapes (strict-load compatible), so the
reference's trainers (compatibility_prediction_trainer.py:59-64) and demo (demo/app.py:102-129)
can use it unchanged.  There is NO PyTorch/CPU fallback: without the HIP library or off a HIP
device the forward raises.

`nn.TransformerEncoder` / `nn.Linear` below are parameter containers (names, shapes and torch's
default initialisation, so the same seed yields the reference's initial weights); their torch
forward is never called.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Type, TypeVar, Union

import numpy as np
import torch
from PIL import Image
from torch import nn

from . import _lib as L
from .configs import OutfitXConfig
from .datatypes import (OutfitCompatibilityPredictionTask, OutfitComplementaryItemRetrievalTask,
                        OutfitFillInTheBlankTask, OutfitPrecomputeEmbeddingTask)
from .encoders import ItemEncoder
from .engine import Engine

_ACT_NAMES = {"mish": "mish", "gelu": "gelu", "relu": None}


def _activation_id(act) -> int:
    name = act if isinstance(act, str) else getattr(act, "__name__", "")
    if name == "mish":
        return L.ACT_MISH
    if name == "gelu":
        return L.ACT_GELU
    raise NotImplementedError(f"transformer activation {act!r} has no fused epilogue (mish / gelu are built)")


import weakref

_LIVE_MODELS: "weakref.WeakSet" = weakref.WeakSet()


def _after_any_optimizer_step(optimizer, args, kwargs):
    """Global torch hook: whichever optimizer stepped may have rewritten an OutfitX's parameters (fused kernels do not bump
    tensor versions), so the packed operand copies of every live model are re-made on its next forward."""
    for m in list(_LIVE_MODELS):
        m.mark_weights_changed()


try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_hook
    _reg_hook(_after_any_optimizer_step)
except ImportError:                                   # older torch: CPTrainer / callers use mark_weights_changed() themselves
    pass


class _CPTrainFn(torch.autograd.Function):
    """CP path with a hand-written backward (libofx_hip.so: ofx_cp_train_fwd / ofx_cp_train_bwd).  Gradients flow to
    the outfit transformer, outfit_token and cp_ffn; the embeddings are data (cp_trainer feeds precomputed ones)."""

    @staticmethod
    def forward(ctx, eng, x, mask, F, drop, *params):
        if isinstance(x, tuple):                  # indexed (varlen) input: (table, item_index, cu_seqlens, max_len)
            table, idx, cu, max_len = x
            logits, tape = eng.cp_train_fwd_indexed(table, idx, cu, max_len, *drop)
            bl = (cu.numel() - 1, max_len)
        else:
            logits, tape = eng.cp_train_fwd(x, mask, *drop)
            bl = (x.shape[0], x.shape[1])
        ctx.eng, ctx.tape, ctx.bl, ctx.F, ctx.drop = eng, tape, bl, F, drop
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.params = params if getattr(eng, "grad_sink", False) else None
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        eng = ctx.eng
        B, Lq = ctx.bl
        dests = _sink_dests(ctx.params, (1, 4)) if ctx.params is not None else None
        if dests is not None:
            eng.train_bwd_into("cp", ctx.tape, dlogits.contiguous(), B, Lq, dests, True, *ctx.drop)
            ctx.tape = None
            return (None,) * (5 + len(ctx.shapes))
        g = eng.cp_train_bwd(ctx.tape, dlogits.contiguous(), B, Lq, *ctx.drop)
        ctx.tape = None
        _, offs = eng.grad_layout()
        out = _grad_views(g, offs, ctx.shapes, eng.desc.d_model, ctx.F, skip=(1, 4))   # target_item_image_emb, cir_ffn: not on the CP path
        return (None, None, None, None, None, *out)


def _sink_dests(params, skip):
    """Gradient-sink mode (OutfitX.grad_sink, set by trainer.CPTrainer): every parameter on the path already owns a dense fp32
    .grad on its own device -> the backward kernels add into those buffers directly."""
    dests = []
    for i, p in enumerate(params):
        if i in skip:
            dests.append(None)
            continue
        g = p.grad
        if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device or not p.requires_grad:
            return None
        dests.append(g)
    return dests


def _grad_views(g, offs, shapes, D, F, skip):
    Fp = (F + 127) // 128 * 128
    out = []
    for i, shp in enumerate(shapes):
        if i in skip:
            out.append(None)
            continue
        k = (i - 5) % 12 if i >= 5 else -1
        if k == 4:
            out.append(g[offs[i]:offs[i] + Fp * D].view(Fp, D)[:F])
        elif k == 5:
            out.append(g[offs[i]:offs[i] + F])
        elif k == 6:
            out.append(g[offs[i]:offs[i] + D * Fp].view(D, Fp)[:, :F])
        else:
            n = 1
            for v in shp:
                n *= v
            out.append(g[offs[i]:offs[i] + n].view(shp))
    return out


class _CIRTrainFn(torch.autograd.Function):
    """CIR / FITB path with the hand-written backward (ofx_cir_train_fwd / ofx_cir_train_bwd): gradients reach the
    transformer, target_item_image_emb and cir_ffn (the CP head and outfit_token are not on this path)."""

    @staticmethod
    def forward(ctx, eng, setin, txt, F, drop, *params):
        y, tape, bl = eng.cir_train_fwd(setin, txt, *drop)
        ctx.eng, ctx.tape, ctx.bl, ctx.F, ctx.drop = eng, tape, bl, F, drop
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.params = params if getattr(eng, "grad_sink", False) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        eng = ctx.eng
        dests = _sink_dests(ctx.params, (0, 2, 3)) if ctx.params is not None else None
        if dests is not None:
            eng.train_bwd_into("cir", ctx.tape, dy.contiguous(), *ctx.bl, dests, True, *ctx.drop)
            ctx.tape = None
            return (None,) * (5 + len(ctx.shapes))
        g = eng.cir_train_bwd(ctx.tape, dy.contiguous(), *ctx.bl, *ctx.drop)
        ctx.tape = None
        _, offs = eng.grad_layout()
        return (None, None, None, None, None, *_grad_views(g, offs, ctx.shapes, eng.desc.d_model, ctx.F, skip=(0, 2, 3)))


class OutfitX(nn.Module):
    Tasks = TypeVar("Tasks", OutfitComplementaryItemRetrievalTask, OutfitCompatibilityPredictionTask,
                    OutfitFillInTheBlankTask, OutfitPrecomputeEmbeddingTask)

    def __init__(self, cfg: Optional[OutfitXConfig] = None, precision: str = "bf16x3", tower_precision: str = L.DEFAULT_TOWER_PRECISION,
                 train_precision: str = "bf16"):
        super().__init__()
        self.cfg = cfg if cfg is not None else OutfitXConfig()
        t = self.cfg.transformer
        self.item_encoder = ItemEncoder(self.cfg.item_encoder)
        d = self.item_encoder.d_embed
        layer = nn.TransformerEncoderLayer(d_model=d, nhead=t.n_head, dim_feedforward=t.d_ffn, dropout=t.dropout,
                                           batch_first=t.batch_first, norm_first=t.norm_first, activation=t.activation)
        self.transformer_encoder = nn.TransformerEncoder(encoder_layer=layer, num_layers=t.n_layers,
                                                         enable_nested_tensor=t.enable_nested_tensor)
        self.outfit_token = nn.Parameter(torch.randn(d) * 0.02, requires_grad=True)
        self.cp_ffn = nn.Sequential(nn.Dropout(t.dropout), nn.Linear(d, 1))
        self.cir_ffn = nn.Sequential(nn.Linear(d, self.cfg.d_embed, bias=False))
        self.target_item_image_emb = nn.Parameter(torch.randn(d // 2) * 0.02, requires_grad=True)
        self.forward_ = {
            OutfitCompatibilityPredictionTask: self._cp_forward,
            OutfitComplementaryItemRetrievalTask: self._cir_forward,
            OutfitFillInTheBlankTask: self._cir_forward,
            OutfitPrecomputeEmbeddingTask: self.precompute_embeddings,
        }
        if not t.norm_first:
            raise NotImplementedError("post-norm encoder layers are outside the scoring path (reference uses norm_first)")
        self.precision = precision
        # Operand format of the set transformer when its input embeddings are produced IN THE SAME CALL by the reduced-precision
        # towers (encoder_input_dict): those embeddings carry the towers' 2^-9 (bf16) / 2^-12 (f16) operand rounding, so the
        # three-product bf16x3 scheme (kept for precomputed fp32 embeddings, where it holds 1e-5) buys nothing there; one f16
        # product (2^-12) is below the bf16 towers' input error and a third of the GEMM work (end-to-end error vs the oracle
        # unchanged at 5-7e-3, step 25.64 -> 24.98 ms; with f16 towers it would double 4e-4 to 8e-4, so it applies to bf16
        # towers only: the default 'f16w2x' towers and plain 'f16' keep bf16x3).  None = always self.precision.
        self.tower_fed_precision: Optional[str] = "f16"
        _LIVE_MODELS.add(self)
        self.train_precision = train_precision      # operand format of the training step (the reference trains under fp16 autocast on CUDA,
                                                    # compatibility_prediction_trainer.py:63; bf16 has the same 8-bit-or-better products and fp32's range)
        self.item_encoder.set_precision(tower_precision)
        self._engines: Dict[Any, Engine] = {}
        # eval-mode `model(task=CP, outfit_embedding=None, ..., encoder_input_dict={'images': device tensor, 'texts': token dict})` calls of a
        # repeated shape are stream-captured on their second occurrence and replayed as ONE graph launch from the third on (graphs.ForwardReplay:
        # same kernels, bit-identical logits, a fresh result tensor per call); False = every call launch by launch
        self.graph_replay = True
        self._replay = None

    # ------------------------------------------------------------------ plumbing
    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    def __getstate__(self):
        s = self.__dict__.copy()
        s["_engines"] = {}
        s["_replay"] = None
        return s

    def mark_weights_changed(self) -> None:
        """Force a re-pack of the operand copies on the next call.  `_engine` notices parameter updates through the tensors'
        version counters, which in-place torch ops bump - but FUSED optimizers (torch.optim.AdamW(fused=True)) write the
        parameters without bumping them.  Every optimizer step therefore also lands here through the global post-step
        hook below, and trainer.CPTrainer calls it explicitly."""
        for eng in self._engines.values():
            eng.signature["outfit"] = None
        if getattr(self, "_replay", None) is not None:
            self._replay.clear()              # captured steps launch on the packed copies of the old weights

    def _outfit_tensors(self) -> List[torch.Tensor]:
        out = [self.outfit_token, self.target_item_image_emb, self.cp_ffn[1].weight, self.cp_ffn[1].bias, self.cir_ffn[0].weight]
        for l in self.transformer_encoder.layers:
            out += [l.self_attn.in_proj_weight, l.self_attn.in_proj_bias, l.self_attn.out_proj.weight, l.self_attn.out_proj.bias,
                    l.linear1.weight, l.linear1.bias, l.linear2.weight, l.linear2.bias,
                    l.norm1.weight, l.norm1.bias, l.norm2.weight, l.norm2.bias]
        return out

    def _engine(self, precision: Optional[str] = None) -> Engine:
        _LIVE_MODELS.add(self)               # (also covers unpickled / deep-copied instances, whose __init__ never ran)
        dev = self.device
        precision = precision or self.precision
        key = (dev, precision)
        eng = self._engines.get(key)
        if eng is None:
            desc = L.default_desc()
            t = self.cfg.transformer
            desc.d_model, desc.n_head, desc.d_ffn, desc.n_layers = self.item_encoder.d_embed, t.n_head, t.d_ffn, t.n_layers
            desc.max_items = min(self.cfg.max_length, 63)
            desc.outfit_act = _activation_id(t.activation)
            desc.ln_eps = self.transformer_encoder.layers[0].norm1.eps
            eng = Engine(dev, desc, precision=precision)
            self._engines[key] = eng
        ts = self._outfit_tensors()
        sig = tuple((p.data_ptr(), p._version) for p in ts)
        if eng.signature["outfit"] != sig:
            eng.pack_outfit(ts)
            eng.signature["outfit"] = sig
        return eng

    # ------------------------------------------------------------------ reference API
    def forward(self, task: Type["OutfitX.Tasks"], *args, **kwargs):
        """outfit_x.py:97-104: dispatch on the task class; unknown key -> KeyError."""
        return self.forward_[task](*args, **kwargs)

    def precompute_embeddings(self, images: List[List[Union[np.ndarray, Image.Image]]], texts: List[List[str]]):
        """outfit_x.py:107-118: one item per 'outfit' -> [B, d_embed]."""
        return self.item_encoder(images, texts)[:, 0, :]

    def encode_items(self, images, texts):
        """north-star alias of ItemEncoder.forward: [B,L] items -> [B,L,d_embed]."""
        return self.item_encoder(images, texts)

    def _run_encoder(self, outfit_embedding, outfit_mask, prefix=None, precision: Optional[str] = None):
        eng = self._engine(precision)
        return eng, eng.set_encoder(outfit_embedding, outfit_mask, prefix)

    def _tower_fed(self) -> Optional[str]:
        """Precision of the set transformer for embeddings the towers produced in this call (see __init__)."""
        bf16_towers = getattr(self.item_encoder.image_enc, "tower_precision", None) == "bf16"
        return self.tower_fed_precision if (self.tower_fed_precision and self.precision == "bf16x3" and bf16_towers) else None

    # ------------------------------------------------------------------ indexed (varlen) input, SURVEY.md §8f N3
    def set_embedding_table(self, table: torch.Tensor) -> None:
        """Keep the precomputed item embeddings [n_items, d_embed] resident on the model's device; batches can then be
        index lists (`item_index`, `cu_seqlens` from processor.OutfitXIndexed*Processor) instead of padded tensors."""
        self.embedding_table = table.to(device=self.device, dtype=torch.float32).contiguous()

    def _indexed(self, item_index, cu_seqlens, embedding_table, max_len):
        table = embedding_table if embedding_table is not None else getattr(self, "embedding_table", None)
        if table is None:
            raise ValueError("indexed input needs an embedding table: pass embedding_table= or call set_embedding_table()")
        return table, item_index, cu_seqlens, int(max_len if max_len is not None else min(self.cfg.max_length, 63 if not (self.training and torch.is_grad_enabled()) else 31))

    def _cp_forward(self, outfit_embedding: Optional[torch.Tensor] = None, outfit_mask: Optional[torch.Tensor] = None,
                    encoder_input_dict: Optional[dict] = None, *, item_index: Optional[torch.Tensor] = None,
                    cu_seqlens: Optional[torch.Tensor] = None, embedding_table: Optional[torch.Tensor] = None,
                    max_len: Optional[int] = None) -> torch.Tensor:
        """outfit_x.py:120-144 -> raw compatibility logits [B,1].  Beyond the reference's arguments: the indexed form
        (item_index, cu_seqlens[, embedding_table, max_len]) — same result as the padded tensors built from those rows."""
        if item_index is not None:
            spec = self._indexed(item_index, cu_seqlens, embedding_table, max_len)
            if self.training and torch.is_grad_enabled():
                return self._cp_train_forward(spec, None)
            eng = self._engine()
            return eng.cp_head(eng.set_encoder_indexed(*spec))
        prec = None
        if encoder_input_dict is not None:
            if outfit_embedding is None and not self.training and getattr(self, "graph_replay", False):
                from .graphs import ForwardReplay
                if ForwardReplay.eligible(self, outfit_mask, encoder_input_dict):
                    if self._replay is None:
                        self._replay = ForwardReplay()
                    return self._replay.run(self, outfit_mask, encoder_input_dict,
                                            lambda: self._cp_eval(outfit_mask, encoder_input_dict["images"], None, encoder_input_dict["texts"]))
            outfit_embedding = self.item_encoder(**encoder_input_dict)
            prec = self._tower_fed()
        if self.training and torch.is_grad_enabled():
            return self._cp_train_forward(outfit_embedding, outfit_mask)
        eng, row0 = self._run_encoder(outfit_embedding, outfit_mask, precision=prec)
        return eng.cp_head(row0)

    def _cp_eval(self, outfit_mask, images, prepared_texts, texts=None) -> torch.Tensor:
        """The eval-mode CP call with the item encoder in it, launch by launch (what graphs.ForwardReplay captures; `prepared_texts` =
        tokens already staged on the device)."""
        with torch.no_grad():
            emb = self.item_encoder(images, texts, prepared_texts=prepared_texts)
            eng, row0 = self._run_encoder(emb, outfit_mask, precision=self._tower_fed())
            return eng.cp_head(row0)

    def sink_ready(self, skip=(1, 4)) -> bool:
        """True when the next CP backward will add straight into the parameters' own .grad buffers (gradient-sink mode with a dense
        fp32 .grad on every tensor of the path): only then are a layer's gradients final where the backward records its event -
        with the autograd fallback they land in a private buffer and AccumulateGrad adds them later (trainer.CPTrainer checks this
        before arming the overlapped reduction)."""
        return bool(getattr(self, "grad_sink", False)) and _sink_dests(self._outfit_tensors(), skip) is not None

    def arm_bwd_layer_events(self, events) -> None:
        """Data-parallel overlap (trainer.CPTrainer): the next backward of the training step records events[l] when layer l's
        gradients are final."""
        self._engine(self.train_precision).arm_layer_events(events)

    def _cp_train_forward(self, outfit_embedding, outfit_mask):
        """CP trainer step (compatibility_prediction_trainer.py:57-81): forward with a tape, backward in libofx_hip.so."""
        t = self.cfg.transformer
        if self.train_precision not in ("bf16", "f16"):
            raise ValueError("train_precision must be 'bf16' or 'f16'")
        if isinstance(outfit_embedding, torch.Tensor) and outfit_embedding.requires_grad:
            raise NotImplementedError("gradients w.r.t. the item embeddings (encoder fine-tuning) are not built")
        eng = self._engine(self.train_precision)
        eng.grad_sink = bool(getattr(self, "grad_sink", False))
        # dropout masks are hash(seed, site, element); the seed is drawn from torch's global generator, so
        # torch.manual_seed() makes a run reproducible (same distribution as torch's dropout, not the same stream)
        p = float(t.dropout)
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p > 0.0 else 0
        self.last_dropout = (p, seed)
        return _CPTrainFn.apply(eng, outfit_embedding, outfit_mask, t.d_ffn, (p, seed), *self._outfit_tensors())

    def _cir_forward(self, outfit_embedding: Optional[torch.Tensor] = None, outfit_mask: Optional[torch.Tensor] = None,
                     target_item_text_embedding: Optional[torch.Tensor] = None, *, item_index: Optional[torch.Tensor] = None,
                     cu_seqlens: Optional[torch.Tensor] = None, embedding_table: Optional[torch.Tensor] = None,
                     max_len: Optional[int] = None) -> torch.Tensor:
        """outfit_x.py:147-172 -> target-item embedding [B, d_embed] (indexed form as in _cp_forward)."""
        if self.training and torch.is_grad_enabled():
            t = self.cfg.transformer
            if self.train_precision not in ("bf16", "f16"):
                raise ValueError("train_precision must be 'bf16' or 'f16'")
            if (isinstance(outfit_embedding, torch.Tensor) and outfit_embedding.requires_grad) or target_item_text_embedding.requires_grad:
                raise NotImplementedError("gradients w.r.t. the item / text embeddings (encoder fine-tuning) are not built")
            setin = (self._indexed(item_index, cu_seqlens, embedding_table, max_len) if item_index is not None
                     else (outfit_embedding, outfit_mask))
            p = float(t.dropout)
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p > 0.0 else 0
            self.last_dropout = (p, seed)
            eng = self._engine(self.train_precision)
            eng.grad_sink = bool(getattr(self, "grad_sink", False))
            return _CIRTrainFn.apply(eng, setin, target_item_text_embedding, t.d_ffn, (p, seed), *self._outfit_tensors())
        eng = self._engine()
        prefix = eng.cir_prefix(target_item_text_embedding)
        if item_index is not None:
            table, idx, cu, ml = self._indexed(item_index, cu_seqlens, embedding_table, max_len)
            row0 = eng.set_encoder_indexed(table, idx, cu, ml, prefix)
        else:
            row0 = eng.set_encoder(outfit_embedding, outfit_mask, prefix)
        return eng.cir_head(row0)
