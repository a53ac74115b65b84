"""Multi-GPU layout of the scoring path: one process per GPU (torchrun), `torch.distributed` with
backend 'nccl' (= RCCL over xGMI on ROCm) or 'gloo' (CPU tests).

* CP / FITB / item encode: outfits are independent -> the batch is split across ranks, weights are
  replicated, and the forward needs NO collective.  `shard_range` gives a rank's
  slice; `gather_scores` is the optional final all-gather of [B/N] logits (4 KB).
* CIR retrieval (BASELINE config 4): the POOL is row-sharded.  Every rank scores all queries
  against its shard (fp32-exact distances + local top-k), then ONE `all_gather_into_tensor` of the
  per-shard (dist, global index) candidate lists, packed into one byte record — nq*k*12 B per rank, 600 KB at 1000x50, latency-bound on
  the fully connected xGMI mesh — and a local merge with a deterministic tie-break (smaller global
  index).  This step has no reference call site: the reference scores CIR on one GPU
  (complementary_item_retrieval_trainer.py:240-249); results are identical to the unsharded call.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) of n units for `rank` (first n % world ranks get one more)."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_scores(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """All-gather per-rank score slices (possibly uneven) back into batch order."""
    world = dist.get_world_size(group)
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(out, sizes)], 0)


def sharded_topk(queries: torch.Tensor, pool_shard: torch.Tensor, k: int, shard_base: int,
                 local_topk: Callable, merge: Callable, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """CIR scoring with a row-sharded pool.
    local_topk(Q, P, k, index_base) -> (idx [nq,k] int64 global, dist [nq,k] f32)   (Engine.l2_topk on the GPU)
    merge(idx_parts [W,nq,k], dist_parts [W,nq,k]) -> (idx [nq,k], dist [nq,k])      (engine.topk_merge on the GPU)
    Every rank returns the same global top-k."""
    world = dist.get_world_size(group)
    idx, dst = local_topk(queries, pool_shard, k, shard_base)
    if world == 1:
        return idx, dst
    # ONE collective: the rank's (global index int64, distance fp32) lists travel as one packed byte record
    # [nq*k*8 B of indices | nq*k*4 B of distances] (600 KB at 1000 x 50; latency-bound on the xGMI mesh, so one launch, not two)
    nb_i, nb_d = idx.numel() * 8, dst.numel() * 4
    rec = torch.empty(nb_i + nb_d, dtype=torch.uint8, device=idx.device)
    rec[:nb_i] = idx.contiguous().view(torch.uint8).reshape(-1)
    rec[nb_i:] = dst.contiguous().view(torch.uint8).reshape(-1)
    if idx.device.type == "cuda" and dist.get_backend(group) == "gloo":
        # gloo has no device all-gather: the record goes through the host (CPU rehearsals of the N > 1 path on a one-GPU box; RCCL takes device memory)
        host = torch.empty(world * (nb_i + nb_d), dtype=torch.uint8)
        dist.all_gather_into_tensor(host, rec.cpu(), group=group)
        flat = host.to(idx.device)
    else:
        flat = torch.empty(world * (nb_i + nb_d), dtype=torch.uint8, device=idx.device)     # 1-D: the form RCCL and gloo both accept
        dist.all_gather_into_tensor(flat, rec, group=group)
    allrec = flat.view(world, nb_i + nb_d)
    idx_all = allrec[:, :nb_i].contiguous().view(torch.int64).view((world,) + tuple(idx.shape))
    dst_all = allrec[:, nb_i:].contiguous().view(torch.float32).view((world,) + tuple(dst.shape))
    return merge(idx_all, dst_all)


def cir_topk(engine, queries: torch.Tensor, pool_shard: torch.Tensor, k: int, shard_base: int, group=None):
    """GPU path: Engine.l2_topk + RCCL all-gather + topk_merge kernel."""
    from .engine import topk_merge
    return sharded_topk(queries, pool_shard, k, shard_base,
                        lambda Q, P, kk, base: engine.l2_topk(Q, P, kk, index_base=base), topk_merge, group)
