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

    def fn():
        return model(task=CP, outfit_embedding=None, outfit_mask=outfit_mask, encoder_input_dict={"images": images, "texts": texts})

    return CapturedCall(fn, images.device, warmup, inputs={"images": images, "texts": texts, "outfit_mask": outfit_mask})
