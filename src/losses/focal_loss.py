from outfitx_amd.losses import FocalLoss  # noqa: F401
