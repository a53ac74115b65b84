"""outfitx_amd — MI355X-native (gfx950) implementation of OutfitX's compatibility-scoring forward
path behind the reference's `src.models` Python API.  See DESIGN.md / INTEGRATION.md."""
from .configs import ItemEncoderConfig, OutfitXConfig, TransformerConfig  # noqa: F401
from .datatypes import (FashionItem, OutfitCompatibilityPredictionTask,  # noqa: F401
                        OutfitComplementaryItemRetrievalTask, OutfitFillInTheBlankTask,
                        OutfitPrecomputeEmbeddingTask)


def __getattr__(name):  # torch-heavy modules are imported on first use
    if name in ("OutfitX",):
        from .outfit_x import OutfitX
        return OutfitX
    if name in ("ItemEncoder", "CLIPImageEncoder", "CLIPTextEncoder"):
        from . import encoders
        return getattr(encoders, name)
    raise AttributeError(name)
