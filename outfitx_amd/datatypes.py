"""Task marker types and FashionItem (reference src/models/datatypes/*.py).  The task CLASSES are
the dispatch keys of OutfitX.forward (outfit_x.py:84-104); instances are what the dataset /
collate code passes around.  Declared with pydantic like upstream so existing pickles and
keyword construction keep working."""
from __future__ import annotations

from typing import List, Optional, Union

import numpy as np
import torch
from PIL import Image
from pydantic import BaseModel, ConfigDict, Field


class FashionItem(BaseModel):
    model_config = ConfigDict(arbitrary_types_allowed=True)
    item_id: Optional[int] = Field(default=None, description="id of the item in the item table")
    category: Optional[str] = Field(default="", description="category string (the text fed to the text tower)")
    image: Optional[Union[Image.Image, torch.Tensor]] = Field(default=None)
    description: Optional[str] = Field(default="")
    metadata: Optional[dict] = Field(default_factory=dict)
    embedding: Optional[np.ndarray] = Field(default=None, description="[img ‖ txt] item embedding")
    text_embedding: Optional[np.ndarray] = Field(default=None, description="category text embedding")


class _OutfitTask(BaseModel):
    outfit: List[FashionItem] = Field(default_factory=list)

    def __len__(self):
        return len(self.outfit)


class OutfitCompatibilityPredictionTask(_OutfitTask):
    pass


class OutfitComplementaryItemRetrievalTask(_OutfitTask):
    target_item: FashionItem = Field(default_factory=FashionItem)


class OutfitFillInTheBlankTask(_OutfitTask):
    target_item: FashionItem = Field(default_factory=FashionItem)


class OutfitPrecomputeEmbeddingTask(BaseModel):
    fashion_item: FashionItem
