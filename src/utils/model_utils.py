from outfitx_amd.encoders import aggregate_embeddings, flatten_seq_to_one_dim, freeze_model  # noqa: F401
