from outfitx_amd.datatypes import (FashionItem, OutfitCompatibilityPredictionTask,  # noqa: F401
                                   OutfitComplementaryItemRetrievalTask, OutfitFillInTheBlankTask,
                                   OutfitPrecomputeEmbeddingTask)
