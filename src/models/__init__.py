from outfitx_amd.outfit_x import OutfitX  # noqa: F401  (reference: src/models/__init__.py:1)
