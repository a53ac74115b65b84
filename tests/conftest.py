import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
W_SEED = 7  # weight seed the golden vectors were generated with (oracle/gen_golden.py)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def rel_err(a, b):
    """max|a-b| / max|b| — the tolerance metric used throughout (stated in DESIGN.md)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture(scope="session")
def ot_weights():
    from outfitx_amd import synth
    return synth.outfit_transformer_weights(W_SEED)


@pytest.fixture(scope="session")
def vit_weights():
    from outfitx_amd import synth
    return synth.vision_weights(W_SEED)


@pytest.fixture(scope="session")
def txt_weights():
    from outfitx_amd import synth
    return synth.text_weights(W_SEED)
