from outfitx_amd.losses import SetWiseRankingLoss  # noqa: F401
