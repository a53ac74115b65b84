"""Collate processors: task objects -> the tensor dict that OutfitX.forward consumes.  They run in
DataLoader worker processes, so they stay picklable and CPU-only (the reference's inline test
checks exactly that: outfit_x_processor_factory.py:38-79).

Reference: src/models/processor/outfit_x/outfit_x_base_processor.py:13-81 and the CP / FITB / CIR /
PE processors next to it.  Same outputs (keys, shapes, dtypes, zero pad rows, mask True on pads);
the padding is one numpy copy per outfit instead of a torch.cat/stack chain per item.
"""
from __future__ import annotations

from typing import List, Literal, Optional, Sequence, Type

import numpy as np
import torch

from .configs import OutfitXConfig
from .datatypes import (OutfitCompatibilityPredictionTask, OutfitComplementaryItemRetrievalTask,
                        OutfitFillInTheBlankTask, OutfitPrecomputeEmbeddingTask)


class OutfitXBaseProcessor:
    def __init__(self, cfg: OutfitXConfig):
        self.cfg = cfg
        self.text_pad = ""
        self.pad_emb = torch.zeros(self.cfg.item_encoder.dim_per_modality * 2)

    def _get_max_length(self, sequences) -> int:
        """base_processor.py:45-55"""
        if self.cfg.padding == "max_length":
            return self.cfg.max_length
        longest = max(len(s) for s in sequences)
        return min(self.cfg.max_length, longest) if self.cfg.truncation else longest

    def _to_tensor_and_padding(self, sequences: Sequence[Sequence[np.ndarray]], pad_value=None):
        """base_processor.py:20-43,57-81 -> (embeddings [B,L,D] float32, mask [B,L] bool, True = pad)."""
        L = self._get_max_length(sequences)
        pad = self.pad_emb if pad_value is None else pad_value
        D = int(pad.shape[-1])
        emb = np.empty((len(sequences), L, D), np.float32)
        emb[:] = pad.numpy() if isinstance(pad, torch.Tensor) else np.asarray(pad, np.float32)
        mask = np.ones((len(sequences), L), bool)
        for i, seq in enumerate(sequences):
            n = min(len(seq), L)
            if n:
                emb[i, :n] = np.asarray(seq[:n], dtype=np.float32)
            mask[i, :n] = False
        return torch.from_numpy(emb), torch.from_numpy(mask)

    def _build_input_dict(self, queries):
        """CIR / FITB input dict (complementary_item_retrieval_processor.py:95-114)."""
        emb, mask = self._to_tensor_and_padding([[it.embedding for it in q.outfit] for q in queries])
        txt = torch.from_numpy(np.stack([np.asarray(q.target_item.text_embedding, np.float32) for q in queries]))
        return {"task": OutfitComplementaryItemRetrievalTask, "outfit_embedding": emb, "outfit_mask": mask,
                "target_item_text_embedding": txt}


class OutfitXCompatibilityPredictionTaskProcessor(OutfitXBaseProcessor):
    """compatibility_prediction_task_processor.py:6-22"""

    def __call__(self, batch):
        queries, labels = zip(*batch)
        emb, mask = self._to_tensor_and_padding([[it.embedding for it in q.outfit] for q in queries])
        return {"input_dict": {"task": OutfitCompatibilityPredictionTask, "outfit_embedding": emb, "outfit_mask": mask},
                "label": torch.tensor(labels, dtype=torch.float)}


class OutfitXFillInTheBlankTaskProcessor(OutfitXBaseProcessor):
    """fill_in_the_blank_task_processor.py:7-40 (emits the CIR task class on purpose, :36)."""

    def __call__(self, batch):
        queries, cands, answers = zip(*batch)
        return {"input_dict": self._build_input_dict(list(queries)),
                "candidate_item_embedding": torch.stack([torch.as_tensor(c) for c in cands]),
                "answer_index": torch.tensor(answers, dtype=torch.long)}


class OutfitXComplementaryItemRetrievalTaskProcessor(OutfitXBaseProcessor):
    """complementary_item_retrieval_processor.py:7-114"""

    def __init__(self, run_mode: Literal["train", "valid", "test"], *args, **kwargs):
        super().__init__(*args, **kwargs)
        if run_mode not in ("train", "valid", "test"):
            raise ValueError(f"unknown run_mode {run_mode!r}")
        self.run_mode = run_mode

    def __call__(self, batch):
        if self.run_mode == "test":
            queries, _ = zip(*batch)
            return {"input_dict": self._build_input_dict(list(queries)),
                    "pos_item_id": [q.target_item.item_id for q in queries]}
        queries, negs = zip(*batch)
        out = {"input_dict": self._build_input_dict(list(queries))}
        if self.run_mode == "valid":
            out["pos_item_id"] = [q.target_item.item_id for q in queries]
        out["pos_item_embedding"] = torch.from_numpy(np.stack([np.asarray(q.target_item.embedding, np.float32) for q in queries]))
        out["neg_items_embedding"], out["neg_items_mask"] = self._to_tensor_and_padding([list(n) for n in negs])
        return out


class OutfitXPrecomputeEmbeddingTaskProcessor(OutfitXBaseProcessor):
    """precompute_embedding_processor.py:7-16"""

    def __call__(self, batch: List[OutfitPrecomputeEmbeddingTask]):
        return {"input_dict": {"task": OutfitPrecomputeEmbeddingTask,
                               "images": [[t.fashion_item.image] for t in batch],
                               "texts": [[t.fashion_item.category] for t in batch]},
                "item_id": [t.fashion_item.item_id for t in batch]}


class OutfitXIndexedProcessor:
    """Index-emitting collate: the item embeddings stay in a device-resident table
    (`OutfitX.set_embedding_table`), a batch is `item_index` int32 [total items] + `cu_seqlens` int32 [B+1] — a few KB
    instead of the reference's padded fp32 [B, 16, 1024] (outfit_x_base_processor.py:20-81).  Outfits are truncated
    to cfg.max_length items exactly like `_get_max_length` does with truncation=True; the encoder result is identical
    to the padded form (padding never reaches the arithmetic: pad-free sets).  Picklable, CPU-only.

    id_to_row: item_id -> row of the table (embedding_store.EmbeddingTable.index)."""

    def __init__(self, task: Type, cfg: Optional[OutfitXConfig] = None, id_to_row=None):
        self.task, self.cfg = task, cfg if cfg is not None else OutfitXConfig()
        self.id_to_row = id_to_row

    def _rows(self, items) -> List[int]:
        ids = [it.item_id for it in items][: self.cfg.max_length]
        return [int(i) for i in ids] if self.id_to_row is None else [int(self.id_to_row[i]) for i in ids]

    def _index(self, outfits):
        rows = [self._rows(o) for o in outfits]
        cu = np.zeros(len(rows) + 1, np.int32)
        cu[1:] = np.cumsum([len(r) for r in rows])
        flat = np.fromiter((i for r in rows for i in r), np.int32, count=int(cu[-1]))
        return torch.from_numpy(flat), torch.from_numpy(cu)

    def __call__(self, batch):
        if self.task is OutfitCompatibilityPredictionTask:
            queries, labels = zip(*batch)
            idx, cu = self._index([q.outfit for q in queries])
            return {"input_dict": {"task": OutfitCompatibilityPredictionTask, "item_index": idx, "cu_seqlens": cu},
                    "label": torch.tensor(labels, dtype=torch.float)}
        if self.task is OutfitFillInTheBlankTask:
            queries, cands, answers = zip(*batch)
            idx, cu = self._index([q.outfit for q in queries])
            txt = torch.from_numpy(np.stack([np.asarray(q.target_item.text_embedding, np.float32) for q in queries]))
            return {"input_dict": {"task": OutfitComplementaryItemRetrievalTask, "item_index": idx, "cu_seqlens": cu,
                                   "target_item_text_embedding": txt},
                    "candidate_item_embedding": torch.stack([torch.as_tensor(c) for c in cands]),
                    "answer_index": torch.tensor(answers, dtype=torch.long)}
        raise ValueError(f"no indexed collate for task {self.task!r}")


class OutfitXProcessorFactory:
    """outfit_x_processor_factory.py:16-36"""

    @staticmethod
    def get_processor(task: Type, cfg: Optional[OutfitXConfig] = None,
                      run_mode: Optional[Literal["train", "valid", "test"]] = None, *args, **kwargs):
        if cfg is None:
            cfg = OutfitXConfig()
        if task is OutfitCompatibilityPredictionTask:
            return OutfitXCompatibilityPredictionTaskProcessor(cfg=cfg)
        if task is OutfitComplementaryItemRetrievalTask:
            if run_mode is None:
                raise ValueError("run_mode must be specified for OutfitComplementaryItemRetrievalTask")
            return OutfitXComplementaryItemRetrievalTaskProcessor(run_mode=run_mode, cfg=cfg)
        if task is OutfitFillInTheBlankTask:
            return OutfitXFillInTheBlankTaskProcessor(cfg=cfg)
        if task is OutfitPrecomputeEmbeddingTask:
            return OutfitXPrecomputeEmbeddingTaskProcessor(cfg=cfg)
        return None
