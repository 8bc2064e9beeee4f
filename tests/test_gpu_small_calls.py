"""The reference's own call pattern -- one point per BallTree::query / query_radius call (benches/ball_tree.rs:22-62).

Small corpora (<= 4096 rows): the whole call is one launch (select.hip, tiny_query_kernel) reading the query from and
writing the answer to mapped pinned memory.  A handful of queries against a large corpus: the batched pipeline with a
workgroup per CU, one scout tile per run and -inf thresholds for the padding queries.  Every answer bit-identical to the
oracle's brute force (the canonical (distance, index) order)."""
import numpy as np
import pytest

from conftest import uniform

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,dim", [(1, 1), (2, 3), (64, 10), (65, 7), (300, 1), (1000, 3), (2500, 16), (4096, 5)])
def test_small_corpus_one_launch_calls(pn, oracle_mod, dtype, n, dim):
    rng = np.random.default_rng(1000 + n + dim)
    pts = rng.random((n, dim)).astype(dtype)
    if n >= 8:
        pts[5] = pts[2]                      # equal rows: ties go by index
        pts[7, 0] = np.nan                   # a NaN row sorts last (ordered-float)
    tree = pn.BallTree.euclidean(pts)
    qs = np.concatenate([pts[:2], rng.random((3, dim)).astype(dtype)])[: max(1, min(5, n + 2))]
    for k in (1, 2, 5, n, n + 3):
        if k < 1:
            continue
        want_i, want_d = oracle_mod.brute_knn(pts, qs, k)
        for a in range(len(qs)):            # one point per call
            i, d = tree.query(qs[a], k)
            assert np.array_equal(i, want_i[a]) and d.tobytes() == want_d[a].tobytes(), (k, a)
        bi, bd = tree.query_batch(qs, k)    # and a few per call
        assert np.array_equal(bi, want_i) and bd.tobytes() == want_d.tobytes(), k
    assert tree.query(qs[0], 0)[0].shape == (0,)
    _, d2 = oracle_mod.brute_knn(pts, qs, min(3, n))
    fin = d2[np.isfinite(d2)]
    radii = [dtype(0), dtype(1e9)] + ([dtype(fin.max() * 1.0000001), dtype(np.median(fin))] if fin.size else [])
    for r in radii + [dtype(np.nan), dtype(-1)]:
        for a in range(len(qs)):
            assert np.array_equal(tree.query_radius(qs[a], r), oracle_mod.brute_radius(pts, qs[a], r)), (float(r), a)
    tree.close()


def test_small_corpus_zip_truncation(pn, oracle_mod):
    """A query shorter or longer than the rows is zipped to the shorter length (src/distance.rs:27-28)."""
    pts = uniform((200, 6), 77, np.float64)
    tree = pn.BallTree.euclidean(pts)
    for qc in (4, 6, 9):
        q = uniform((qc,), 78 + qc, np.float64)
        m = min(qc, 6)
        want_i, want_d = oracle_mod.brute_knn(np.ascontiguousarray(pts[:, :m]), q[None, :m], 5)
        i, d = tree.query(q, 5)
        assert np.array_equal(i, want_i[0]) and d.tobytes() == want_d[0].tobytes(), qc


@pytest.mark.parametrize("nq", [1, 5, 33, 300])
def test_a_handful_of_queries_against_a_large_corpus(pn, oracle_mod, nq):
    n, dim, k = 300_000, 128, 10
    pts, qs = uniform((n, dim), 4201), uniform((nq, dim), 4202 + nq)
    tree = pn.BallTree.euclidean(pts)
    want_i, want_d = oracle_mod.brute_knn(pts, qs, k)
    for eng in ("auto", "bf16"):
        tree.set_engine(eng)
        tree.stats(reset=True)
        idx, dist = tree.query_batch(qs, k)
        assert np.array_equal(idx, want_i) and dist.tobytes() == want_d.tobytes(), eng
        assert tree.stats()["fallback_queries"] <= 1
    i1, d1 = tree.query(qs[0], k)  # BallTree::query(point, k)
    assert np.array_equal(i1, want_i[0]) and d1.tobytes() == want_d[0].tobytes()
