from outfitx_amd.processor import OutfitXProcessorFactory  # noqa: F401  (reference: src/models/processor/__init__.py:1)
