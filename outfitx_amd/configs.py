"""Configuration dataclasses of the OutfitX scoring path, field-for-field compatible with the
reference's (src/models/configs/{item_encoder,outfit_x,transformer}_config.py) so existing callers
can construct and pickle them unchanged.  Quirks that callers can observe are kept on purpose:
`TransformerConfig.batch_first/norm_first` are 1-tuples (trailing commas upstream,
transformer_config.py:20-21), `OutfitXConfig.d_embed` is always overwritten with
2*dim_per_modality (outfit_x_config.py:23) and the default encoder type is 'slip'
(item_encoder_config.py:9) even though only 'clip' is built here.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Literal, Union

import torch.nn.functional as F
from torch import Tensor

_ENCODERS = {
    # type: (attribute holding the checkpoint name, checkpoint, per-modality width)
    "clip": ("clip_model_name", "patrickjohncyh/fashion-clip", 512),
    "resnet_hf_sentence_bert": ("text_model_name", "sentence-transformers/all-MiniLM-L6-v2", 64),
    "slip": ("slip_model_name", "hf-hub:Marqo/marqo-fashionSigLIP", 768),
}


@dataclass
class ItemEncoderConfig:
    type: Literal["clip", "resnet_hf_sentence_bert", "slip"] = "slip"
    norm_out: bool = True
    aggregation_method: Literal["concat", "sum", "mean"] = "concat"

    def __post_init__(self):
        if self.type not in _ENCODERS:
            raise ValueError(f"Unsupported type: {self.type}")
        attr, name, width = _ENCODERS[self.type]
        setattr(self, attr, name)
        self.dim_per_modality: int = width


@dataclass
class TransformerConfig:
    n_head: int = 16
    d_ffn: int = 2024
    n_layers: int = 6
    dropout: float = 0.3
    norm_out: bool = False
    batch_first: bool = (True,)   # sic: a 1-tuple upstream; truthy
    norm_first: bool = (True,)    # sic
    activation: Union[str, Callable[[Tensor], Tensor]] = F.mish
    enable_nested_tensor: bool = False


@dataclass
class OutfitXConfig:
    padding: Literal["longest", "max_length"] = "max_length"
    max_length: int = 16
    truncation: bool = True
    d_embed: int = 1024
    item_encoder: ItemEncoderConfig = field(default_factory=ItemEncoderConfig)
    transformer: TransformerConfig = field(default_factory=TransformerConfig)

    def __post_init__(self):
        self.d_embed = 2 * self.item_encoder.dim_per_modality
        attr = _ENCODERS[self.item_encoder.type][0]
        self.model_name = getattr(self.item_encoder, attr).split("/")[-1]
