"""A forward of fixed shape as ONE HIP graph launch.

Every entry point of libofx_hip.so launches on the stream it is given and none synchronises (include/ofx.h), so a whole scoring call - both
towers on their two streams, the fuser, the set transformer, the head: ~250 kernels and two small H2D copies for BASELINE configs[1] - can
be stream-captured once and replayed with one launch per call.  What that buys: the host issues one launch instead of ~250 (1.5-2 ms of Python +
ctypes per 33 ms step), the step no longer depends on the host keeping the queue fed (on a busy host the launch-by-launch step shows idle gaps
of tens of ms whenever the issuing thread is late), and the small bubble at every back-pressure wait of the runtime goes (33.24 -> 33.00 ms per
step on a quiet box, per-step times flat).  The logits are bit-identical to the launch-by-launch call (tests/test_gpu_model.py).

Constraints, all the usual ones of a captured graph:
  * shapes, dtypes and the ADDRESSES of every input are frozen: feed a new batch by copying it into the captured tensors (`CapturedCall.inputs`);
    pinned host tensors that the call copies to the device (token ids / masks) are re-read at every replay, so overwrite them in place;
  * host-side decisions are frozen too: for texts that is the longest token row of the captured batch (the tower computes that many positions):
    a replayed batch must not hold a longer one;
  * the returned tensors are owned by the graph and overwritten by the next replay: clone what must survive;
  * inference only (no autograd), parameters must not be re-packed (an optimizer step invalidates the capture).
"""
from __future__ import annotations

from typing import Any, Callable, Optional

import torch


class CapturedCall:
    """`fn()` (no arguments: it closes over its input tensors) captured into a HIP graph after `warmup` eager calls on the capture stream.
    `replay()` launches the graph on torch's current stream and returns fn's (static) result."""

    def __init__(self, fn: Callable[[], Any], device: torch.device, warmup: int = 2, inputs: Optional[dict] = None):
        if device.type != "cuda":
            raise ValueError("CapturedCall needs a HIP device")
        self.inputs = inputs or {}
        self.device = device
        self.check: Optional[Callable[[], None]] = None
        self._graph = torch.cuda.CUDAGraph()
        cur = torch.cuda.current_stream(device)
        cap = torch.cuda.Stream(device)
        cap.wait_stream(cur)
        with torch.cuda.stream(cap), torch.no_grad():
            for _ in range(max(1, warmup)):          # one-time attribute calls, allocator pool of the capture stream, weight packs
                fn()
        cur.wait_stream(cap)
        torch.cuda.synchronize(device)
        # thread_local: only this thread's calls are checked against the capture (a collective library's watchdog thread polling its events must not void it)
        with torch.no_grad(), torch.cuda.graph(self._graph, stream=cap, capture_error_mode="thread_local"):
            self.output = fn()
        torch.cuda.synchronize(device)
        self.replays = 0

    def replay(self):
        if self.check is not None:
            self.check()              # frozen host-side decisions (capture_cp_forward: the longest token row) still hold, or raise
        self._graph.replay()
        self.replays += 1
        return self.output


def capture_cp_forward(model, outfit_mask: torch.Tensor, images: torch.Tensor, texts: dict, warmup: int = 2) -> CapturedCall:
    """The reference's `model(task=OutfitCompatibilityPredictionTask, outfit_embedding=None, outfit_mask=..., encoder_input_dict={'images', 'texts'})`
    (outfit_x.py:120-144 with the item encoder in the call) as a CapturedCall.  images: device tensor [B, n, 3, 224, 224]; texts: {'input_ids',
    'attention_mask'} [B, n, T] int64, pinned host tensors (re-read at each replay) or device tensors."""
    from .datatypes import OutfitCompatibilityPredictionTask as CP
    if model.training:
        raise ValueError("capture_cp_forward: put the model in eval() mode (the training step is not capturable: its parameters change)")
    if not isinstance(images, torch.Tensor) or images.device.type != "cuda":
        raise ValueError("capture_cp_forward: images must be a device tensor (host preprocessing cannot be captured)")
    for k, v in texts.items():
        if isinstance(v, torch.Tensor) and v.device.type == "cpu" and not v.is_pinned():
            raise ValueError(f"capture_cp_forward: texts['{k}'] lives in pageable host memory; pin it (or move it to the device) - a pageable copy is synchronous and cannot be captured")

    enc = model.item_encoder.text_enc
    if enc.dedup_texts or enc.cache_texts:
        raise ValueError("capture_cp_forward: turn text_enc.dedup_texts / cache_texts off - their row maps are computed on the host per batch and would be frozen into the graph")

    def fn():
        return model(task=CP, outfit_embedding=None, outfit_mask=outfit_mask, encoder_input_dict={"images": images, "texts": texts})

    was = getattr(model, "graph_replay", False)
    model.graph_replay = False                     # an explicit capture: the calls inside it are the launch-by-launch ones
    try:
        cc = CapturedCall(fn, images.device, warmup, inputs={"images": images, "texts": texts, "outfit_mask": outfit_mask})
    finally:
        model.graph_replay = was
    ids = texts["input_ids"]
    if ids.device.type == "cpu":                   # host ids: the tower computes max(EOS position) + 1 token positions, decided on the host at capture time
        flat = ids.reshape(-1, ids.shape[-1])
        cc.captured_tokens = max(1, min(flat.shape[1], max(enc._lengths(flat))))

        def check():
            now = max(enc._lengths(flat))          # the pinned tensor is re-read at every replay: what it holds NOW must fit what was captured
            if now > cc.captured_tokens:
                raise ValueError(f"replay: a token row of {now} positions, but the graph was captured for {cc.captured_tokens}: capture again (a longer row would be silently truncated)")
        cc.check = check
    return cc


class ForwardReplay:
    """What gives the drop-in call `model(task=CP, outfit_embedding=None, outfit_mask=..., encoder_input_dict={'images', 'texts'})` the
    speed of a captured graph without the caller doing anything (OutfitX._cp_forward, eval mode): the SECOND call with the same key runs
    launch by launch like the first and is then captured; from the third call on the step is one hipGraphLaunch.

    key = address / shape / dtype of the image tensor (the graph reads the caller's own tensor: no 1.2 GB copy per call; a caller that
    passes another tensor at another address starts another entry), shapes of the token tensors and the outfit mask, the number of token
    positions the text tower computes (host ids: longest row; device ids: all T), the operand schemes, stream options, the library's
    ofx_tune generation.  Token ids / attention mask / outfit mask are COPIED into entry-owned device tensors on the caller's stream
    before each replay (1 MB; pinned or device sources asynchronously, pageable ones synchronously - exactly how the launch-by-launch
    call stages them), so the graph itself never reads caller-owned host memory.  An entry dies when any parameter changes (data_ptr / version
    signature, or OutfitX.mark_weights_changed from an optimizer step) - the packed operand copies it launches on would be stale.
    The result is a fresh tensor per call (a clone of the graph's static output), as the reference returns.  At most `capacity` entries
    (each owns its graph's workspace pool: ~3 GB at 256 outfits x 8 items)."""

    def __init__(self, capacity: int = 2, capture_after: int = 2):
        self.capacity, self.capture_after = capacity, capture_after
        self.entries: dict = {}
        self.seen: dict = {}
        self.failed: set = set()          # keys whose capture raised: never tried again (the launch-by-launch call keeps serving them)
        self.stats = {"eager": 0, "captures": 0, "replays": 0, "capture_failures": 0}

    def clear(self) -> None:
        self.entries.clear()
        self.seen.clear()
        self.failed.clear()

    @staticmethod
    def eligible(model, outfit_mask, enc_in) -> bool:
        if not isinstance(enc_in, dict) or set(enc_in) - {"images", "texts"}:
            return False
        images, texts = enc_in.get("images"), enc_in.get("texts")
        if not (isinstance(images, torch.Tensor) and images.device.type == "cuda" and images.is_contiguous() and images.dim() == 5):
            return False
        if not (isinstance(texts, dict) and isinstance(texts.get("input_ids"), torch.Tensor) and texts["input_ids"].dim() == 3):
            return False
        att = texts.get("attention_mask")
        if att is not None and not (isinstance(att, torch.Tensor) and att.shape == texts["input_ids"].shape):
            return False
        if not (isinstance(outfit_mask, torch.Tensor) and outfit_mask.dim() == 2):
            return False
        enc = model.item_encoder
        if enc.text_enc.dedup_texts or enc.text_enc.cache_texts or enc.cfg.aggregation_method != "concat":
            return False
        return not torch.cuda.is_current_stream_capturing()

    def run(self, model, outfit_mask, enc_in, eager: Callable[[], torch.Tensor]):
        """-> logits.  `eager` is the launch-by-launch call (used until the key has been seen capture_after times, and whenever capture fails)."""
        from . import _lib as L
        lib = L.load()
        if lib.ofx_profile_enabled():
            return eager()
        images, texts = enc_in["images"], enc_in["texts"]
        ids, att = texts["input_ids"], texts.get("attention_mask")
        enc = model.item_encoder
        T = ids.shape[-1]
        tokens = T
        if ids.device.type == "cpu":
            tokens = max(1, min(T, max(enc.text_enc._lengths(ids.reshape(-1, T)))))
        key = (images.data_ptr(), tuple(images.shape), images.dtype, tuple(ids.shape), att is not None, tokens, tuple(outfit_mask.shape),
               model.precision, model.tower_fed_precision, enc.image_enc.tower_precision, enc.text_enc.tower_precision, enc.overlap_towers, enc.side_stream_priority,
               enc.image_enc.vit_streams, lib.ofx_config_generation(), torch.cuda.current_stream(images.device).cuda_stream)
        sig = tuple((p.data_ptr(), p._version) for p in model.parameters())
        e = self.entries.get(key)
        if e is not None and e["sig"] != sig:
            del self.entries[key]
            e = None
        if e is None and key in self.failed:
            self.stats["eager"] += 1
            return eager()
        if e is None:
            n = self.seen.get(key, 0) + 1
            if len(self.seen) > 64:
                self.seen.clear()
            self.seen[key] = n
            if n < self.capture_after:
                self.stats["eager"] += 1
                return eager()
            out = eager()                                  # this call's result; the capture below serves the following ones
            self.stats["eager"] += 1
            e = self._capture(model, outfit_mask, images, ids, att, tokens, sig)
            if e is None:
                if len(self.failed) < 256:
                    self.failed.add(key)
            else:
                while len(self.entries) >= self.capacity:
                    self.entries.pop(next(iter(self.entries)))
                self.entries[key] = e
            return out
        dev = images.device
        e["ids"].copy_(ids.reshape(e["ids"].shape), non_blocking=True)
        if att is not None:
            e["att"].copy_(att.reshape(e["att"].shape), non_blocking=True)
        e["mask"].copy_(outfit_mask, non_blocking=True)
        e["graph"].replay()
        self.stats["replays"] += 1
        return e["out"].clone()

    def _capture(self, model, outfit_mask, images, ids, att, tokens, sig):
        dev = images.device
        n_rows, T = ids.numel() // ids.shape[-1], ids.shape[-1]
        ids_d = torch.empty(n_rows, T, dtype=torch.int64, device=dev); ids_d.copy_(ids.reshape(n_rows, T))
        att_d = None
        if att is not None:
            att_d = torch.empty(n_rows, T, dtype=torch.int64, device=dev); att_d.copy_(att.reshape(n_rows, T))
        mask_d = outfit_mask.to(dev).clone()
        lengths = None if ids.device.type != "cpu" else [tokens] * n_rows     # only their maximum (positions computed) is a host-side decision
        prepared = (ids_d, att_d, lengths, ids.shape[0], None)

        def fn():
            return model._cp_eval(mask_d, images, prepared)

        try:
            cc = CapturedCall(fn, dev, warmup=1)
        except Exception:                                  # the launch-by-launch call stays the one that runs
            self.stats["capture_failures"] += 1
            return None
        self.stats["captures"] += 1
        return {"graph": cc, "out": cc.output, "ids": ids_d, "att": att_d, "mask": mask_d, "sig": sig}
