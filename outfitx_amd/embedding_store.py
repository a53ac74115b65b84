"""Precomputed item-embedding store ("next" row N4).

The reference's precompute script dumps one pickle per rank, `{'ids': list[int], 'embeddings': ndarray[N,1024] fp32}`
named "<model_name>_embedding_subset_<rank>.pkl" (src/trains/trainers/precompute_embedding_script.py:47-55), and the
trainers load and concatenate them (compatibility_prediction_trainer.py:329-349).  This module reads that layout and
offers an mmap-able variant (`.npy` embeddings + `.ids.npy`) that avoids unpickling 1024-wide rows into Python
objects: batches are then gathered by index straight from the memory map (and pinned for the H2D copy).
"""
from __future__ import annotations

import glob
import os
import pickle
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np


def shard_name(model_name: str, rank: int) -> str:
    return f"{model_name}_embedding_subset_{rank}.pkl"


def save_pickle_shard(directory: str, model_name: str, rank: int, ids: Sequence[int], embeddings: np.ndarray) -> str:
    """Write one shard in the reference's layout."""
    emb = np.ascontiguousarray(embeddings, dtype=np.float32)
    if emb.ndim != 2 or emb.shape[0] != len(ids):
        raise ValueError("embeddings must be [len(ids), d]")
    path = os.path.join(directory, shard_name(model_name, rank))
    with open(path, "wb") as f:
        pickle.dump({"ids": [int(i) for i in ids], "embeddings": emb}, f)
    return path


def load_pickle_shards(directory: str, model_name: str) -> Tuple[np.ndarray, np.ndarray]:
    """All shards of a model, concatenated in rank order -> (ids int64 [N], embeddings fp32 [N,d])."""
    paths = sorted(glob.glob(os.path.join(directory, f"{model_name}_embedding_subset_*.pkl")),
                   key=lambda p: int(p.rsplit("_", 1)[1].split(".")[0]))
    if not paths:
        raise FileNotFoundError(f"no embedding shards for '{model_name}' in {directory}")
    ids: List[int] = []
    embs = []
    for p in paths:
        with open(p, "rb") as f:
            d = pickle.load(f)
        ids.extend(d["ids"])
        embs.append(np.asarray(d["embeddings"], np.float32))
    return np.asarray(ids, np.int64), np.concatenate(embs, 0)


def convert_to_mmap(directory: str, model_name: str) -> str:
    """Pickle shards -> '<model_name>_embeddings.npy' (+ '.ids.npy'), readable with mmap_mode='r'."""
    ids, emb = load_pickle_shards(directory, model_name)
    base = os.path.join(directory, f"{model_name}_embeddings")
    np.save(base + ".npy", emb)
    np.save(base + ".ids.npy", ids)
    return base + ".npy"


class EmbeddingTable:
    """id -> row lookup over an mmap-ed (or in-memory) [N,d] fp32 table."""

    def __init__(self, ids: np.ndarray, embeddings: np.ndarray):
        self.ids = np.asarray(ids, np.int64)
        self.embeddings = embeddings
        order = np.argsort(self.ids, kind="stable")
        self._sorted_ids, self._order = self.ids[order], order

    @classmethod
    def open(cls, directory: str, model_name: str) -> "EmbeddingTable":
        base = os.path.join(directory, f"{model_name}_embeddings")
        if os.path.exists(base + ".npy"):
            return cls(np.load(base + ".ids.npy"), np.load(base + ".npy", mmap_mode="r"))
        return cls(*load_pickle_shards(directory, model_name))

    def rows(self, item_ids: Iterable[int]) -> np.ndarray:
        q = np.asarray(list(item_ids), np.int64)
        pos = np.searchsorted(self._sorted_ids, q)
        if (pos >= len(self._sorted_ids)).any() or (self._sorted_ids[np.minimum(pos, len(self._sorted_ids) - 1)] != q).any():
            raise KeyError("unknown item id in lookup")
        return self._order[pos]

    def gather(self, item_ids: Iterable[int]) -> np.ndarray:
        """[len(ids), d] fp32 (a copy; sorted row access for the memory map)."""
        r = self.rows(item_ids)
        o = np.argsort(r, kind="stable")
        out = np.empty((len(r), self.embeddings.shape[1]), np.float32)
        out[o] = self.embeddings[r[o]]
        return out

    def index(self) -> Dict[int, int]:
        """item_id -> row of `embeddings` (what processor.OutfitXIndexedProcessor needs to emit table rows)."""
        return {int(i): k for k, i in enumerate(self.ids)}

    def as_dict(self) -> Dict[int, np.ndarray]:
        """The reference's in-memory form (id -> row view)."""
        return {int(i): self.embeddings[k] for k, i in enumerate(self.ids)}
