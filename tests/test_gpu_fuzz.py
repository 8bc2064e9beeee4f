"""Randomised parity sweep (tests/fuzz_knn.py): the auto engine with all its tiers against the oracle's brute
force, bit-exact, over random shapes and data families (uniform, centred, tight clusters, duplicates, sorted rows)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_shapes_match_the_oracle(pn, oracle_mod, seed):
    import fuzz_knn
    rng = np.random.default_rng(seed)
    for c in range(6):
        assert fuzz_knn.run_case(100 * seed + c, rng)


@pytest.mark.parametrize("seed", [21, 22])
def test_random_sharded_handles_match_the_oracle(pn, oracle_mod, seed):
    """tests/fuzz_sharded.py: virtual shards on one GPU, f32 / f64, Euclidean / Cosine, k-NN and radius."""
    import fuzz_sharded
    rng = np.random.default_rng(seed)
    for c in range(6):
        assert fuzz_sharded.run_case(100 * seed + c, rng)


@pytest.mark.parametrize("seed", [31])
def test_random_large_indexes_with_the_seed_model(pn, oracle_mod, seed):
    """tests/fuzz_seed_model.py: 10^5 .. 6 10^5 rows (where the seed model is fitted), ten data families, four kinds of
    queries, f32 / f64, Euclidean / Cosine: the bf16 tier with the model on and off == the exact engine == the oracle."""
    import fuzz_seed_model
    rng = np.random.default_rng(seed)
    for c in range(5):
        assert fuzz_seed_model.run_case(100 * seed + c, rng)
