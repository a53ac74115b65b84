"""Import-path shim: `from src.models import OutfitX` etc. resolve to outfitx_amd, so callers written
against the reference (trainers, demo) pick up the MI355X implementation without edits."""
