from outfitx_amd.encoders import CLIPImageEncoder, CLIPTextEncoder, ItemEncoder  # noqa: F401
