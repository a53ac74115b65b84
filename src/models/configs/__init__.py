from outfitx_amd.configs import ItemEncoderConfig, OutfitXConfig, TransformerConfig  # noqa: F401
