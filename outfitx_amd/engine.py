"""Thin Python host over the C ABI: owns an ofx_handle, packs torch parameters into it, lends
torch-allocated workspaces and output tensors, and launches on torch's current HIP stream.

PyTorch is plumbing here (device memory, streams); all arithmetic happens in libofx_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib as L


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _f32c(t: torch.Tensor, device) -> torch.Tensor:
    if t.device != device or t.dtype != torch.float32 or not t.is_contiguous():
        t = t.to(device=device, dtype=torch.float32).contiguous()
    return t


class Engine:
    """One per (module, device).  `precision`: MFMA operand format of the outfit transformer
    ('bf16x3' | 'bf16' | 'f16'); `tower_precision`: operand scheme of the CLIP towers - a name `_lib.tower_scheme` accepts: 'f16w2x' (default:
    every ViT weight split; all 100 weight seeds swept at the bench's batch size inside 1e-3 of the reference, worst 6.3e-4) | 'f16w2h' (the qkv
    correction on ViT layers 0-5 only: 3 % faster, worst 7.35e-4 at that batch size, outside 1e-3 on one small-logit 8-outfit draw) | 'f16w2' (faster; worst seeds at 1.0e-3) | 'f16x3' (every
    tower GEMM three-product) | 'f16' | 'bf16' (single product, outside 1e-3) | any of them + '@qkv=<layers>;fc1=...' (per-layer rungs)."""

    def __init__(self, device: torch.device, desc: Optional[L.ModelDesc] = None,
                 precision: str = "bf16x3", tower_precision: str = L.DEFAULT_TOWER_PRECISION):
        self.lib = L.load()
        if device.type != "cuda":
            raise L.OfxError(f"outfitx_amd runs on an MI355X HIP device only (got {device}); there is no CPU path")
        self.device = torch.device("cuda", device.index if device.index is not None else torch.cuda.current_device())
        d = desc if desc is not None else L.default_desc()
        d.outfit_precision = L.PRECISIONS[precision]
        for k, v in L.tower_scheme(tower_precision).items():       # ValueError on an unknown scheme
            setattr(d, k, v)
        self.desc = d
        self.h = self.lib.ofx_create(self.device.index, C.byref(d))
        if not self.h:
            raise L.OfxError("ofx_create failed: " + self.lib.ofx_last_error().decode())
        self._ws: Dict[int, torch.Tensor] = {}
        self._keep: List[torch.Tensor] = []
        self.signature = {"outfit": None, "vision": None, "text": None}

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.ofx_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---------------------------------------------------------------- weights
    def _pack(self, fn, tensors: Sequence[torch.Tensor], what: str):
        ts = [_f32c(t.detach(), self.device) for t in tensors]
        arr = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        with torch.cuda.device(self.device):
            L.check(fn(self.h, arr, len(ts), _stream(self.device)), f"ofx_pack_{what}_weights")
        # the pack kernels read `ts` asynchronously on the current stream; keep temporaries alive until they are done
        self._keep = [t for t in ts]

    def pack_outfit(self, tensors): self._pack(self.lib.ofx_pack_outfit_weights, tensors, "outfit")
    def pack_vision(self, tensors): self._pack(self.lib.ofx_pack_vision_weights, tensors, "vision")
    def pack_text(self, tensors): self._pack(self.lib.ofx_pack_text_weights, tensors, "text")

    # ---------------------------------------------------------------- workspace
    def workspace(self, nbytes: int) -> torch.Tensor:
        key = _stream(self.device)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(int(nbytes * 1.05) + 4096, dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def ws_bytes(self, op: int, n: int, length: int) -> int:
        return int(self.lib.ofx_workspace_bytes(self.h, op, n, length))

    # ---------------------------------------------------------------- outfit transformer
    def set_encoder(self, x: torch.Tensor, mask: torch.Tensor, prefix: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x [B,L,D] float, mask [B,L] bool (True = pad) -> encoder output at the prefix token [B,D] fp32."""
        B, Lq, D = x.shape
        x = _f32c(x, self.device)
        m = mask.to(device=self.device)
        m = (m if m.dtype == torch.bool else m != 0).contiguous().view(torch.uint8)
        stride = 0
        if prefix is not None:
            prefix = _f32c(prefix, self.device)
            stride = 0 if prefix.dim() == 1 else D
        out = torch.empty(B, D, dtype=torch.float32, device=self.device)
        nb = self.ws_bytes(L.OP_SET_ENCODER, B, Lq)
        ws = self.workspace(nb)
        with torch.cuda.device(self.device):
            L.check(self.lib.ofx_set_encoder_fwd(self.h, _ptr(x), _ptr(m), _ptr(prefix), stride, B, Lq, _ptr(out),
                                                 _ptr(ws), ws.numel(), _stream(self.device)), "ofx_set_encoder_fwd")
        return out

    def _index_args(self, table: torch.Tensor, item_index: torch.Tensor, cu_seqlens: torch.Tensor):
        """Validate / stage the indexed (varlen) set input.  Host-side index tensors are checked here (range, monotone
        offsets, items per outfit); device-side ones are trusted (the kernel clamps indices)."""
        if table.device != self.device or table.dtype != torch.float32 or table.dim() != 2 or table.stride(1) != 1:
            raise L.OfxError("embedding table must be a [n, D] fp32 tensor on the engine's device with unit column stride")
        if table.shape[1] != self.desc.d_model:
            raise L.OfxError(f"embedding table has {table.shape[1]} columns, the model needs {self.desc.d_model}")
        B = cu_seqlens.numel() - 1
        if cu_seqlens.device.type == "cpu":
            cu = cu_seqlens.to(torch.int64)
            n = cu[1:] - cu[:-1]
            if B < 1 or int(cu[0]) != 0 or bool((n < 0).any()) or int(cu[-1]) != item_index.numel():
                raise ValueError("cu_seqlens must start at 0, be non-decreasing and end at len(item_index)")
            max_items = int(n.max())
        else:
            max_items = None
        if item_index.device.type == "cpu" and item_index.numel():
            lo, hi = int(item_index.min()), int(item_index.max())
            if lo < 0 or hi >= table.shape[0]:
                raise IndexError(f"item_index out of range [0, {table.shape[0]}): min {lo}, max {hi}")
        idx = item_index.to(device=self.device, dtype=torch.int32, non_blocking=True).contiguous()
        cud = cu_seqlens.to(device=self.device, dtype=torch.int32, non_blocking=True).contiguous()
        return idx, cud, B, max_items

    def set_encoder_indexed(self, table: torch.Tensor, item_index: torch.Tensor, cu_seqlens: torch.Tensor, max_len: int,
                            prefix: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Outfits as row indices into a device-resident embedding table -> encoder output at the prefix token [B,D]."""
        idx, cud, B, max_items = self._index_args(table, item_index, cu_seqlens)
        if max_items is not None and max_items > max_len:
            raise ValueError(f"an outfit holds {max_items} items, max_len is {max_len} (truncate in the processor)")
        D = self.desc.d_model
        stride = 0
        if prefix is not None:
            prefix = _f32c(prefix, self.device)
            stride = 0 if prefix.dim() == 1 else D
        out = torch.empty(B, D, dtype=torch.float32, device=self.device)
        ws = self.workspace(self.ws_bytes(L.OP_SET_ENCODER, B, max_len))
        with torch.cuda.device(self.device):
            L.check(self.lib.ofx_set_encoder_fwd_indexed(self.h, _ptr(table), table.stride(0), table.shape[0], _ptr(idx), _ptr(cud), _ptr(prefix), stride,
                                                         B, max_len, _ptr(out), _ptr(ws), ws.numel(), _stream(self.device)), "ofx_set_encoder_fwd_indexed")
        self._keep_idx = (idx, cud)
        return out

    def cp_train_fwd_indexed(self, table, item_index, cu_seqlens, max_len: int, dropout_p: float = 0.0, seed: int = 0):
        idx, cud, B, max_items = self._index_args(table, item_index, cu_seqlens)
        if max_items is not None and max_items > max_len:
            raise ValueError(f"an outfit holds {max_items} items, max_len is {max_len} (truncate in the processor)")
        tape = torch.empty(int(self.lib.ofx_cp_train_tape_bytes(self.h, B, max_len)), dtype=torch.uint8, device=self.device)
        ws = self.workspace(int(self.lib.ofx_cp_train_ws_bytes(self.h, B, max_len)))
        logits = torch.empty(B, 1, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            L.check(self.lib.ofx_cp_train_fwd_indexed(self.h, _ptr(table), table.stride(0), table.shape[0], _ptr(idx), _ptr(cud), B, max_len, _ptr(logits),
                                                      _ptr(tape), tape.numel(), _ptr(ws), ws.numel(), float(dropout_p), int(seed) & 0xFFFFFFFF,
                                                      _stream(self.device)), "ofx_cp_train_fwd_indexed")
        self._keep_idx = (idx, cud)
        return logits, tape

    def cir_train_fwd(self, setin, target_text: torch.Tensor, dropout_p: float = 0.0, seed: int = 0):
        """Tape-saving CIR forward.  setin: (x [B,L,D], mask [B,L]) or (table, item_index, cu_seqlens, max_len).
        -> (y [B,D] fp32, tape, (B, L))"""
        txt = _f32c(target_text, self.device)
        null = None
        if len(setin) == 4:
            table, item_index, cu_seqlens, Lq = setin
            idx, cud, B, max_items = self._index_args(table, item_index, cu_seqlens)
            if max_items is not None and max_items > Lq:
                raise ValueError(f"an outfit holds {max_items} items, max_len is {Lq} (truncate in the processor)")
            args = (null, null, _ptr(table), table.stride(0), table.shape[0], _ptr(idx), _ptr(cud))
            self._keep_idx = (idx, cud)
        else:
            x, mask = setin
            B, Lq, _ = x.shape
            x = _f32c(x, self.device)
            m = mask.to(device=self.device)
            m = (m if m.dtype == torch.bool else m != 0).contiguous().view(torch.uint8)
            args = (_ptr(x), _ptr(m), null, 0, 0, null, null)
            self._keep_idx = (x, m)
        tape = torch.empty(int(self.lib.ofx_cp_train_tape_bytes(self.h, B, Lq)), dtype=torch.uint8, device=self.device)
        ws = self.workspace(int(self.lib.ofx_cp_train_ws_bytes(self.h, B, Lq)))
        y = torch.empty(B, self.desc.d_model, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            L.check(self.lib.ofx_cir_train_fwd(self.h, *args, _ptr(txt), B, Lq, _ptr(y), _ptr(tape), tape.numel(), _ptr(ws), ws.numel(),
                                               float(dropout_p), int(seed) & 0xFFFFFFFF, _stream(self.device)), "ofx_cir_train_fwd")
        return y, tape, (B, Lq)

    def cir_train_bwd(self, tape: torch.Tensor, dy: torch.Tensor, B: int, Lq: int, dropout_p: float = 0.0, seed: int = 0) -> torch.Tensor:
        total, _ = self.grad_layout()
        g = torch.empty(total, dtype=torch.float32, device=self.device)
        d = _f32c(dy, self.device)
        ws = self.workspace(int(self.lib.ofx_cp_train_ws_bytes(self.h, B, Lq)))
        with torch.cuda.device(self.device):
            L.check(self.lib.ofx_cir_train_bwd(self.h, _ptr(tape), tape.numel(), _ptr(d), B, Lq, _ptr(g), total, _ptr(ws), ws.numel(),
                                               float(dropout_p), int(seed) & 0xFFFFFFFF, _stream(self.device)), "ofx_cir_train_bwd")
        return g

    def cp_head(self, row0: torch.Tensor) -> torch.Tensor:
        B = row0.shape[0]
        out = torch.empty(B, 1, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            L.check(self.lib.ofx_cp_head(self.h, _ptr(row0), B, _ptr(out), _stream(self.device)), "ofx_cp_head")
        return out

    def cir_head(self, row0: torch.Tensor) -> torch.Tensor:
        B, D = row0.shape
        out = torch.empty(B, D, dtype=torch.float32, device=self.device)
        ws = self.workspace(B * D * 2 * 3 + 256)
        with torch.cuda.device(self.device):
            L.check(self.lib.ofx_cir_head(self.h, _ptr(row0), B, _ptr(out), _ptr(ws), ws.numel(), _stream(self.device)), "ofx_cir_head")
        return out

    def cir_prefix(self, target_text: torch.Tensor) -> torch.Tensor:
        t = _f32c(target_text, self.device)
        B = t.shape[0]
        out = torch.empty(B, self.desc.d_model, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            L.check(self.lib.ofx_cir_prefix(self.h, _ptr(t), B, _ptr(out), _stream(self.device)), "ofx_cir_prefix")
        return out

    # ---------------------------------------------------------------- training step (CP path, next row N1)
    def grad_layout(self):
        n = 5 + 12 * self.desc.n_layers
        offs = (C.c_size_t * n)()
        total = int(self.lib.ofx_cp_train_grad_floats(self.h, offs, n))
        return total, list(offs)

    def cp_train_fwd(self, x: torch.Tensor, mask: torch.Tensor, dropout_p: float = 0.0, seed: int = 0):
        """Tape-saving CP forward: x [B,L,D], mask [B,L] (True = pad) -> (logits [B,1] fp32, tape uint8 tensor)."""
        B, Lq, D = x.shape
        x = _f32c(x, self.device)
        m = mask.to(device=self.device)
        m = (m if m.dtype == torch.bool else m != 0).contiguous().view(torch.uint8)
        tape = torch.empty(int(self.lib.ofx_cp_train_tape_bytes(self.h, B, Lq)), dtype=torch.uint8, device=self.device)
        ws = self.workspace(int(self.lib.ofx_cp_train_ws_bytes(self.h, B, Lq)))
        logits = torch.empty(B, 1, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            L.check(self.lib.ofx_cp_train_fwd(self.h, _ptr(x), _ptr(m), B, Lq, _ptr(logits), _ptr(tape), tape.numel(),
                                              _ptr(ws), ws.numel(), float(dropout_p), int(seed) & 0xFFFFFFFF, _stream(self.device)), "ofx_cp_train_fwd")
        return logits, tape

    def cp_train_bwd(self, tape: torch.Tensor, dlogits: torch.Tensor, B: int, Lq: int, dropout_p: float = 0.0, seed: int = 0) -> torch.Tensor:
        """d loss / d logits [B,1] -> flat fp32 gradient buffer (layout: grad_layout())."""
        total, _ = self.grad_layout()
        g = torch.empty(total, dtype=torch.float32, device=self.device)
        dl = _f32c(dlogits, self.device)
        ws = self.workspace(int(self.lib.ofx_cp_train_ws_bytes(self.h, B, Lq)))
        with torch.cuda.device(self.device):
            L.check(self.lib.ofx_cp_train_bwd(self.h, _ptr(tape), tape.numel(), _ptr(dl), B, Lq, _ptr(g), total,
                                              _ptr(ws), ws.numel(), float(dropout_p), int(seed) & 0xFFFFFFFF, _stream(self.device)), "ofx_cp_train_bwd")
        return g

    def arm_layer_events(self, events) -> None:
        """events: one torch.cuda.Event per outfit-transformer layer (each recorded at least once, so that its HIP handle exists);
        the NEXT backward call records events[l] when layer l's gradients are final (ofx_train_arm_layer_events)."""
        arr = (C.c_void_p * len(events))(*[int(e.cuda_event) for e in events])
        L.check(self.lib.ofx_train_arm_layer_events(self.h, arr, len(events)), "ofx_train_arm_layer_events")

    def train_bwd_into(self, head: str, tape: torch.Tensor, dhead: torch.Tensor, B: int, Lq: int, dests, accumulate: bool,
                       dropout_p: float = 0.0, seed: int = 0) -> None:
        """Backward of the CP ('cp') or CIR ('cir') path writing every gradient straight into `dests` (one fp32 contiguous
        tensor of the parameter's shape per packed tensor, None for tensors off the path) - p.grad += g without the 75
        accumulate kernels of autograd."""
        ptrs = (C.c_void_p * len(dests))(*[None if t is None else t.data_ptr() for t in dests])
        d = _f32c(dhead, self.device)
        ws = self.workspace(int(self.lib.ofx_cp_train_ws_bytes(self.h, B, Lq)))
        fn = self.lib.ofx_cp_train_bwd_into if head == "cp" else self.lib.ofx_cir_train_bwd_into
        with torch.cuda.device(self.device):
            L.check(fn(self.h, _ptr(tape), tape.numel(), _ptr(d), B, Lq, ptrs, len(dests), int(accumulate), _ptr(ws), ws.numel(),
                       float(dropout_p), int(seed) & 0xFFFFFFFF, _stream(self.device)), f"ofx_{head}_train_bwd_into")

    # ---------------------------------------------------------------- towers
    def vit(self, pixels: torch.Tensor, out: torch.Tensor, col: int, normalize: bool) -> None:
        """pixels [N,3,H,W] fp32 pixel_values -> out[:, col:col+512] (out is [N, ld] fp32 contiguous)."""
        px = _f32c(pixels, self.device)
        N = px.shape[0]
        nb = self.ws_bytes(L.OP_VIT, N, 0)
        ws = self.workspace(nb)
        with torch.cuda.device(self.device):
            L.check(self.lib.ofx_vit_b32_fwd(self.h, _ptr(px), N, _ptr(out), out.stride(0), col, int(normalize),
                                             _ptr(ws), ws.numel(), _stream(self.device)), "ofx_vit_b32_fwd")

    def _stage_images(self, images: Sequence):
        """uint8 images -> (device byte buffer, host offsets / heights / widths): one pinned staging copy, one H2D."""
        import numpy as np
        arrs = []
        for a in images:
            a = np.asarray(a)
            if a.dtype != np.uint8 or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] != 3):
                raise ValueError(f"clip_preprocess takes uint8 [H,W,3] or [H,W] images, got {a.dtype} {a.shape}")
            arrs.append(np.repeat(a[:, :, None], 3, 2) if a.ndim == 2 else a)
        N = len(arrs)
        hs = np.asarray([a.shape[0] for a in arrs], np.int32); ws_ = np.asarray([a.shape[1] for a in arrs], np.int32)
        nbytes = hs.astype(np.int64) * ws_ * 3
        offs = np.zeros(N, np.int64); offs[1:] = np.cumsum((nbytes[:-1] + 15) // 16 * 16)
        total = int(offs[-1] + nbytes[-1])
        stage = getattr(self, "_px_stage", None)
        if stage is None or stage.numel() < total + 4:
            stage = torch.empty(int(total * 1.25) + 4096, dtype=torch.uint8).pin_memory()
            self._px_stage = stage
        elif getattr(self, "_px_event", None) is not None:
            self._px_event.synchronize()      # only the previous batch's H2D copy (not the kernels queued behind it) must have left the staging buffer
        buf = stage.numpy()
        for a, o, n in zip(arrs, offs, nbytes):
            buf[o:o + n] = a.reshape(-1)
        src = stage[:total + 4].to(self.device, non_blocking=True)    # + 4: RGB pixels are fetched as unaligned dwords
        self._px_event = torch.cuda.Event()
        self._px_event.record(torch.cuda.current_stream(self.device))
        self._keep_px = src
        return src, offs, hs, ws_

    def clip_preprocess(self, images: Sequence, size: int, mean: Sequence[float], std: Sequence[float]) -> torch.Tensor:
        """uint8 images ([H,W,3] or [H,W] numpy arrays, any sizes) -> normalised pixel_values [N,3,size,size] fp32 on the
        device: the packed bytes travel once (0.27 MB per 300x300 image instead of 0.6 MB of fp32 pixels) and the resize /
        crop / normalise run on the GPU, bit-identical to PIL + CLIPImageProcessor (ofx_clip_preprocess)."""
        src, offs, hs, ws_ = self._stage_images(images)
        N = len(hs)
        out = torch.empty(N, 3, size, size, dtype=torch.float32, device=self.device)
        I = C.POINTER(C.c_int); LL = C.POINTER(C.c_longlong)
        nb = int(self.lib.ofx_clip_preprocess_ws(hs.ctypes.data_as(I), ws_.ctypes.data_as(I), N, 3, size))
        ws = self.workspace(nb)
        m = (C.c_float * 3)(*mean); sd = (C.c_float * 3)(*std)
        with torch.cuda.device(self.device):
            L.check(self.lib.ofx_clip_preprocess(_ptr(src), offs.ctypes.data_as(LL), hs.ctypes.data_as(I), ws_.ctypes.data_as(I), N, 3, size,
                                                 m, sd, _ptr(out), _ptr(ws), ws.numel(), _stream(self.device)), "ofx_clip_preprocess")
        return out

    def vit_u8(self, images: Sequence, mean: Sequence[float], std: Sequence[float], out: torch.Tensor, col: int, normalize: bool) -> None:
        """uint8 images -> out[:, col:col+512]: the GPU preprocessor feeds the patch-embedding GEMM directly
        (ofx_vit_b32_fwd_u8); same result as clip_preprocess + vit without the fp32 pixel tensor in between."""
        src, offs, hs, ws_ = self._stage_images(images)
        N = len(hs)
        I = C.POINTER(C.c_int); LL = C.POINTER(C.c_longlong)
        nb = int(self.lib.ofx_vit_b32_u8_ws_bytes(self.h, hs.ctypes.data_as(I), ws_.ctypes.data_as(I), N, 3))
        if nb == 0:
            raise ValueError("ofx_vit_b32_u8_ws_bytes: bad image geometry")
        ws = self.workspace(nb)
        m = (C.c_float * 3)(*mean); sd = (C.c_float * 3)(*std)
        with torch.cuda.device(self.device):
            L.check(self.lib.ofx_vit_b32_fwd_u8(self.h, _ptr(src), offs.ctypes.data_as(LL), hs.ctypes.data_as(I), ws_.ctypes.data_as(I), N, 3,
                                                m, sd, _ptr(out), out.stride(0), col, int(normalize), _ptr(ws), ws.numel(),
                                                _stream(self.device)), "ofx_vit_b32_fwd_u8")

    def stage_tokens(self, ids: torch.Tensor, att: Optional[torch.Tensor]):
        """Token ids / mask -> device int64 (non-blocking from pinned memory).  Call this BEFORE enqueuing a long
        kernel sequence: a blocking H2D copy in the middle of a step stalls the host behind everything queued."""
        def mv(t):
            if t is None:
                return None
            if t.device == self.device and t.dtype == torch.int64 and t.is_contiguous():
                return t
            return t.to(device=self.device, dtype=torch.int64, non_blocking=t.device.type == "cpu" and t.is_pinned()).contiguous()
        return mv(ids), mv(att)

    def text(self, ids: torch.Tensor, att: Optional[torch.Tensor], out: torch.Tensor, col: int, normalize: bool,
             lengths: Optional[Sequence[int]] = None) -> None:
        """ids/att [N,T] int64 -> out[:, col:col+512].  lengths: host ints (EOS position + 1) or None."""
        ids_d, att_d = self.stage_tokens(ids, att)
        N, T = ids_d.shape
        Tc = T if lengths is None else max(1, min(T, max(int(v) for v in lengths)))
        arr = None if lengths is None else (C.c_int * N)(*[int(v) for v in lengths])
        nb = self.ws_bytes(L.OP_TEXT, N, Tc)
        ws = self.workspace(nb)
        with torch.cuda.device(self.device):
            L.check(self.lib.ofx_clip_text_fwd(self.h, _ptr(ids_d), _ptr(att_d), arr, N, T, _ptr(out), out.stride(0), col,
                                               int(normalize), _ptr(ws), ws.numel(), _stream(self.device)), "ofx_clip_text_fwd")
        self._keep_ids = (ids_d, att_d)

    # ---------------------------------------------------------------- scoring
    def l2_topk(self, Q: torch.Tensor, P: torch.Tensor, k: int, index_base: int = 0):
        Q = _f32c(Q, self.device); P = _f32c(P, self.device)
        nq, D = Q.shape
        npool = P.shape[0]
        idx = torch.empty(nq, k, dtype=torch.int64, device=self.device)
        dist = torch.empty(nq, k, dtype=torch.float32, device=self.device)
        ws = self.workspace(self.ws_bytes(L.OP_TOPK, nq, npool))
        with torch.cuda.device(self.device):
            L.check(self.lib.ofx_l2_topk(self.h, _ptr(Q), _ptr(P), nq, npool, D, k, index_base, _ptr(idx), _ptr(dist),
                                         _ptr(ws), ws.numel(), _stream(self.device)), "ofx_l2_topk")
        return idx, dist


def fitb_argmin(y_hat: torch.Tensor, cand: torch.Tensor, return_dist: bool = False):
    """torch.cdist(y[B,1,D], cand[B,C,D]).squeeze(1).argmin(-1) on the GPU, fp32, first minimum."""
    lib = L.load()
    dev = y_hat.device
    if dev.type != "cuda":
        raise L.OfxError("fitb_argmin needs HIP tensors; there is no CPU path")
    y = _f32c(y_hat, dev); c = _f32c(cand, dev)
    B, Cn, D = c.shape
    idx = torch.empty(B, dtype=torch.int64, device=dev)
    dist = torch.empty(B, Cn, dtype=torch.float32, device=dev) if return_dist else None
    with torch.cuda.device(dev):
        L.check(lib.ofx_fitb_argmin(_ptr(y), _ptr(c), B, Cn, D, _ptr(idx), _ptr(dist), _stream(dev)), "ofx_fitb_argmin")
    return (idx, dist) if return_dist else idx


def dropout_mask(p: float, seed: int, site: int, rows: int, cols: int, device) -> torch.Tensor:
    """keep-mask / (1 - p) of one dropout site exactly as the training kernels compute it (test aid)."""
    lib = L.load()
    out = torch.empty(rows, cols, dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        L.check(lib.ofx_dropout_mask(float(p), int(seed) & 0xFFFFFFFF, site, rows, cols, _ptr(out), _stream(out.device)), "ofx_dropout_mask")
    return out


FOCAL_REDUCTIONS = {"none": 0, "mean": 1, "sum": 2}


def focal_loss(logits: torch.Tensor, labels: torch.Tensor, alpha: float, gamma: float, upstream: float = 1.0, need_grad: bool = True,
               reduction: str = "mean"):
    """FocalLoss(alpha, gamma, reduction) (src/losses/focal_loss.py:23-41) and upstream * d loss / d logits, one kernel.
    'mean' / 'sum' return a 0-d loss; 'none' returns the [B] unreduced losses (dlogits is then the per-element derivative)."""
    lib = L.load()
    dev = logits.device
    if dev.type != "cuda":
        raise L.OfxError("focal_loss needs HIP tensors; there is no CPU path")
    y = _f32c(logits.reshape(-1), dev); t = _f32c(labels.reshape(-1), dev)
    red = FOCAL_REDUCTIONS[reduction]
    loss = torch.empty((), dtype=torch.float32, device=dev)
    per = torch.empty_like(y) if red == 0 else None
    dl = torch.empty_like(y) if need_grad else None
    with torch.cuda.device(dev):
        L.check(lib.ofx_focal_loss_ex(_ptr(y), _ptr(t), y.numel(), float(alpha), float(gamma), float(upstream), red, _ptr(loss), _ptr(per),
                                      _ptr(dl), _stream(dev)), "ofx_focal_loss_ex")
    return (per if red == 0 else loss), dl


def topk_merge(idx_parts: torch.Tensor, dist_parts: torch.Tensor):
    """[parts,nq,k] per-shard candidates (global indices) -> ([nq,k] idx, [nq,k] dist), ascending, ties -> smaller idx."""
    lib = L.load()
    dev = idx_parts.device
    parts, nq, k = idx_parts.shape
    ii = idx_parts.contiguous(); dd = dist_parts.contiguous()
    idx = torch.empty(nq, k, dtype=torch.int64, device=dev)
    dist = torch.empty(nq, k, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        L.check(lib.ofx_topk_merge(_ptr(ii), _ptr(dd), parts, nq, k, _ptr(idx), _ptr(dist), _stream(dev)), "ofx_topk_merge")
    return idx, dist
