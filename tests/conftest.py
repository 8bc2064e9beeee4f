"""Shared pytest fixtures.

* ``-m "not gpu"``: oracle vs the reference's golden vectors, host logic,
  C-ABI symbol checks, gloo world_size-2 sharding tests.  Runs without a GPU.
* ``-m gpu``: parity tests proper; they call the HIP path through the C ABI and
  compare with the oracle.  Nothing here reads /root/reference at run time.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


HAS_GPU = None


def pytest_collection_modifyitems(config, items):
    global HAS_GPU
    if HAS_GPU is None:
        HAS_GPU = _has_gpu()
    if HAS_GPU:
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def kats():
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def pn():
    """The product package (HIP path). Fails loudly if the extension is missing."""
    import petal_neighbors_amd
    return petal_neighbors_amd


def uniform(shape, seed, dtype=np.float32):
    """Seeded uniform [0,1) with 24 random bits (same generator family as bench.py)."""
    import oracle
    n = int(np.prod(shape))
    return oracle.fill_uniform(n, seed, 0, dtype).reshape(shape)
