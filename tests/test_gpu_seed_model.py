"""Seed model of the bf16 tier (round 4): k-NN calls with small k on a Euclidean index whose model was
accepted at build take their starting thresholds from per-dimension moments of the corpus instead of a scout launch.

A threshold never decides an answer -- the re-rank's proof does, and an unproven query goes to the next tier -- so the
tests hold every answer against the exact engine (itself held against the oracle by tests/test_gpu_parity.py) with the
model on and off, on the corpora the model accepts (uniform-like), on the ones it must refuse (clustered) and under
queries the model has never seen the like of.
"""
import numpy as np
import pytest

from conftest import uniform

pytestmark = pytest.mark.gpu

PN_OPT_SEED_MODEL = 10
PN_OPT_SHARED_THRESHOLDS = 8


def _exact(pn, tree, qs, k):
    tree.set_engine("exact")
    out = tree.query_batch(qs, k)
    tree.set_engine("auto")
    return out


def _same(a, b):
    return np.array_equal(a[0], b[0]) and a[1].tobytes() == b[1].tobytes()


@pytest.mark.parametrize("dtype,dim", [(np.float32, 128), (np.float32, 64), (np.float64, 96)])
def test_model_accepted_on_uniform_rows_answers_identical_on_and_off(pn, dtype, dim):
    n, nq = 400_000, 4096
    pts = uniform((n, dim), 4100 + dim, dtype)
    qs = uniform((nq, dim), 4101 + dim, dtype)
    tree = pn.BallTree.euclidean(pts)
    assert tree.bf16_eligible
    assert tree.seed_model, "uniform rows: the calibration queries' z must agree"
    for k in (1, 10, 32, 100):
        want = _exact(pn, tree, qs, k)
        tree.set_engine("bf16")
        tree.set_option(PN_OPT_SEED_MODEL, 1)
        tree.stats(reset=True)
        on = tree.query_batch(qs, k)
        st_on = tree.stats(reset=True)
        tree.set_option(PN_OPT_SEED_MODEL, 0)
        off = tree.query_batch(qs, k)
        st_off = tree.stats(reset=True)
        assert _same(on, want), (k, "seed model on")
        assert _same(off, want), (k, "seed model off")
        # the model's thresholds do the scout's job: few unproven queries, a comparable number of candidates
        assert st_on["fallback_queries"] <= max(4, nq // 256), (k, st_on)
        assert st_on["candidates"] <= 4 * st_off["candidates"] + 64 * nq, (k, st_on, st_off)
    tree.close()


def test_model_seeds_with_an_explicit_sharing_rank(pn):
    """by default a plan seeded by the model does not share thresholds between segments (they start where sharing would
    only bring them); an explicit PN_OPT_SHARED_THRESHOLDS rank keeps the refreshers on top of the model's seeds --
    including ranks that cut below the k-th neighbour: answers never change"""
    n, dim, nq, k = 1_200_000, 128, 3000, 10
    pts, qs = uniform((n, dim), 3101, np.float32), uniform((nq, dim), 3102, np.float32)
    tree = pn.BallTree.euclidean(pts)
    assert tree.seed_model
    want = _exact(pn, tree, qs, k)
    seen = {}
    for rank in (1, 44, 70, 8):
        if rank == 8:
            tree.close()
            tree = pn.BallTree.euclidean(pts)  # (the unproven queries of rank 8 would switch things off for later calls)
        tree.set_engine("bf16")
        tree.set_option(PN_OPT_SHARED_THRESHOLDS, rank)
        tree.stats(reset=True)
        for _ in range(2):
            got = tree.query_batch(qs, k)
        st = tree.stats(reset=True)
        seen[rank] = (st["fallback_queries"], round(st["candidates"] / st["queries"], 1))
        assert _same(got, want), (rank, seen)
    print("model seeds + sharing rank: (unproven, candidates per query)", seen)
    assert seen[44][1] < seen[1][1], seen   # the refreshers did lower thresholds below the model's seeds
    assert seen[8][0] > nq // 2, seen       # and an absurd rank sends most queries to the next tier
    tree.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_model_on_a_cosine_index(pn, dtype):
    """the model is fitted over the rows normalised in f64 (what the tier's images hold); Cosine answers identical with it
    on and off, and to the exact scan"""
    n, dim, nq = 300_000, 128, 2048
    pts = uniform((n, dim), 4300, dtype) - dtype(0.3)
    qs = uniform((nq, dim), 4301, dtype) - dtype(0.3)
    tree = pn.BallTree.new(pts, pn.distance.Cosine())
    assert tree.bf16_eligible and tree.seed_model
    want = _exact(pn, tree, qs, 10)
    tree.set_engine("bf16")
    tree.stats(reset=True)
    on = tree.query_batch(qs, 10)
    st_on = tree.stats(reset=True)
    tree.set_option(PN_OPT_SEED_MODEL, 0)
    off = tree.query_batch(qs, 10)
    assert _same(on, want) and _same(off, want)
    assert st_on["fallback_queries"] <= nq // 64, st_on
    tree.close()


def test_model_refused_on_clustered_rows(pn):
    rng = np.random.default_rng(77)
    n, dim = 300_000, 64
    centres = rng.normal(size=(40, dim)).astype(np.float32) * 4
    scale = rng.uniform(0.02, 1.0, size=40).astype(np.float32)
    lab = rng.integers(0, 40, size=n)
    pts = centres[lab] + rng.normal(size=(n, dim)).astype(np.float32) * scale[lab, None]
    qs = pts[rng.integers(0, n, size=1024)] + rng.normal(size=(1024, dim)).astype(np.float32) * 0.01
    tree = pn.BallTree.euclidean(pts)
    assert tree.bf16_eligible
    assert not tree.seed_model, "clusters of different widths: no single z fits, the scout launch stays"
    want = _exact(pn, tree, qs, 10)
    tree.set_engine("bf16")
    assert _same(tree.query_batch(qs, 10), want)
    tree.close()


def test_gaussian_rows_whatever_the_build_decided(pn):
    """normal coordinates with per-dimension scales (skewed bound distribution): accepted or not, answers are exact and,
    where the model is used, it leaves few queries to the next tier"""
    rng = np.random.default_rng(78)
    n, dim, nq = 300_000, 100, 2048
    scales = rng.uniform(0.2, 2.0, size=dim).astype(np.float32)
    pts = rng.normal(size=(n, dim)).astype(np.float32) * scales + np.float32(0.7)
    qs = rng.normal(size=(nq, dim)).astype(np.float32) * scales + np.float32(0.7)
    tree = pn.BallTree.euclidean(pts)
    want = _exact(pn, tree, qs, 10)
    tree.set_engine("bf16")
    tree.stats(reset=True)
    assert _same(tree.query_batch(qs, 10), want)
    st = tree.stats(reset=True)
    if tree.seed_model:
        assert st["fallback_queries"] <= nq // 64, st
    tree.close()


def test_model_meets_queries_unlike_the_corpus(pn):
    """accepted on uniform rows, then asked about queries far outside, on top of rows, and all alike: answers stay exact;
    a batch the model serves badly switches it off for the handle (sticky), which the next call shows"""
    n, dim = 400_000, 128
    pts = uniform((n, dim), 4200, np.float32)
    tree = pn.BallTree.euclidean(pts)
    assert tree.seed_model
    rng = np.random.default_rng(5)
    far = uniform((1024, dim), 4201, np.float32) * np.float32(3.0) + np.float32(1.0)
    on_rows = pts[rng.integers(0, n, size=1024)].copy()
    alike = np.repeat(pts[7:8] * np.float32(0.999), 1024, axis=0)
    corner = np.full((1024, dim), 0.5, np.float32) + rng.normal(size=(1024, dim)).astype(np.float32) * np.float32(0.01)
    for name, qs in (("far", far), ("on_rows", on_rows), ("alike", alike), ("corner", corner)):
        want = _exact(pn, tree, qs, 10)
        tree.set_engine("bf16")
        got = tree.query_batch(qs, 10)
        assert _same(got, want), name
        got = tree.query_batch(qs, 10)  # (possibly after the switch-off)
        assert _same(got, want), name + " again"
    tree.close()


def test_model_on_wide_rows(pn):
    """D = 768 (the K-chunked kernel): the model replaces the wide scout launch; answers identical on and off"""
    n, dim, nq = 200_000, 768, 1024
    pts = uniform((n, dim), 4400, np.float32)
    qs = uniform((nq, dim), 4401, np.float32)
    tree = pn.BallTree.euclidean(pts)
    assert tree.bf16_eligible and tree.seed_model
    want = _exact(pn, tree, qs, 10)
    tree.set_engine("bf16")
    tree.stats(reset=True)
    on = tree.query_batch(qs, 10)
    st_on = tree.stats(reset=True)
    tree.set_option(PN_OPT_SEED_MODEL, 0)
    off = tree.query_batch(qs, 10)
    assert _same(on, want) and _same(off, want)
    assert st_on["fallback_queries"] <= nq // 64, st_on
    tree.close()


def test_feedback_state_machine(pn):
    """what finished calls teach a handle about its model (pn_debug_seed_model_feedback feeds observations exactly as a
    finished call does): a few unproven queries -> aim higher; many -> back to the scout; 64 scouted calls later the
    model gets another try, aiming higher, three times at most"""
    import ctypes as C
    from petal_neighbors_amd import _lib
    pts = uniform((120_000, 32), 4500, np.float32)
    tree = pn.BallTree.euclidean(pts)
    assert tree.seed_model
    out = (C.c_int32 * 4)()

    def feed(model, unproven, nq=10_000):
        assert _lib.lib().pn_debug_seed_model_feedback(tree._h, int(model), unproven, nq, out) == 0
        return tuple(out)

    assert feed(True, 0) == (0, 0, 0, 0)
    assert feed(True, 9) == (0, 0, 0, 0)            # 9 of 10^4: below one in 1024
    assert feed(True, 12) == (0, 1, 0, 0)           # a few: aim 1.5x higher
    assert feed(True, 200) == (1, 1, 0, 0)          # more than one in 128: back to the scout
    for i in range(63):
        assert feed(False, 0) == (1, 1, i + 1, 0)   # scouted calls are counted ...
    assert feed(False, 0) == (0, 2, 0, 1)           # ... and the 64th gives the model another try, aiming higher
    for retry in (2, 3):
        assert feed(True, 500)[0] == 1
        for _ in range(64):
            st = feed(False, 0)
        assert st[0] == 0 and st[3] == retry
    assert feed(True, 500)[0] == 1                  # the third retry failed as well:
    for _ in range(200):
        st = feed(False, 0)
    assert st[0] == 1 and st[3] == 3                # it stays off
    # widening alone also ends in the scout: four steps, the fifth switches off
    t2 = pn.BallTree.euclidean(pts)
    seen = [_lib.lib().pn_debug_seed_model_feedback(t2._h, 1, 12, 10_000, out) or tuple(out) for _ in range(5)]
    assert [s[1] for s in seen] == [1, 2, 3, 4, 5] and [s[0] for s in seen] == [0, 0, 0, 0, 1]
    # and the answers of a handle in any of these states stay the exact engine's
    qs = uniform((512, 32), 4501, np.float32)
    for t in (tree, t2):
        want = _exact(pn, t, qs, 10)
        t.set_engine("bf16")
        assert _same(t.query_batch(qs, 10), want)
        t.close()


def test_model_per_shard_of_a_sharded_handle(pn):
    """three virtual shards of 110 k rows each build their own model; PN_OPT_SEED_MODEL is forwarded to every shard;
    answers equal the unsharded index's with the option on and off"""
    from petal_neighbors_amd.sharded import ShardedIndex
    n, dim, nq = 330_000, 32, 1500
    pts = uniform((n, dim), 4600, np.float32)
    qs = uniform((nq, dim), 4601, np.float32)
    tree = pn.BallTree.euclidean(pts)
    want = _exact(pn, tree, qs, 10)
    sh = ShardedIndex.from_host(pts, [0, 0, 0])
    for v in (1, 0):
        sh.set_option(PN_OPT_SEED_MODEL, v)
        gi, gd = sh.query_batch(qs, 10)
        assert np.array_equal(np.asarray(gi, dtype=np.uint64), want[0]) and np.asarray(gd).tobytes() == want[1].tobytes(), v
    sh.close()
    tree.close()
