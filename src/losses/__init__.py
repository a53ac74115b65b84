"""Reference import path `src.losses` (src/losses/__init__.py) -> outfitx_amd."""
from outfitx_amd.losses import FocalLoss, SetWiseRankingLoss  # noqa: F401
