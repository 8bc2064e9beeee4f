"""Pins the CPU oracle against the reference's own golden vectors (CPU only).

Every vector in tests/golden/reference_kats.json is an input + asserted output
transcribed from petal-neighbors' unit tests / doc-tests (file:line in the JSON).
Both oracle implementations are checked: the faithful ball tree (what the
reference runs) and the canonical brute force (the result specification).
"""
import math

import numpy as np
import pytest


def _pts(v):
    if "shape" in v:
        return np.zeros(v["shape"], dtype=np.float64)
    a = np.array(v["points"], dtype=np.float64)
    if v.get("fortran"):
        a = np.asfortranarray(a)  # array.reversed_axes() layout of src/ball_tree.rs:636
    return a


def _close(a, b, tol):
    return abs(a - b) <= tol


def test_all_reference_vectors(oracle_mod, kats):
    o = oracle_mod
    n_ops = 0
    for v in kats["vectors"]:
        pts = _pts(v)
        tree = None
        for op in v["ops"]:
            n_ops += 1
            kind = op["op"]
            where = f'{v["id"]} ({v["cite"]}) {kind}'
            if kind == "build_error":
                with pytest.raises(o.OracleArrayError) as ei:
                    o.Tree(pts)
                assert str(ei.value) == kats["error_strings"][op["expect"]], where
                continue
            if kind == "cosine":  # distance == rdistance for Cosine (src/distance.rs:109-112)
                dt = np.float32 if op.get("dtype") == "f32" else np.float64
                got = o.cosine(np.array(op["a"], dtype=dt), np.array(op["b"], dtype=dt))
                assert _close(got, op["expect"], op["tol"]), where
                continue
            if kind == "pairwise":
                got = o.pairwise(pts)
                assert np.array_equal(got, np.array(op["expect"])), where  # assert_eq! in the reference
                continue
            if kind == "node_init":
                cen, rad = o.node_init(pts, op["idx"])
                assert np.array_equal(cen, np.array(op["expect_centroid"])), where
                assert _close(rad, op["expect_radius"], op["tol"]), where
                continue
            if kind == "max_spread_column":
                assert o.max_spread_column(pts, op["idx"]) == op["expect"], where
                continue
            if kind == "halve_node_indices":
                col = np.array(op["col"], dtype=np.float64)
                idx = o.halve_node_indices(op["idx"], col)
                if "expect_exact" in op:
                    assert list(idx) == op["expect_exact"], where
                else:  # the reference's own assertions, src/ball_tree.rs:821-834
                    assert idx[0] < idx[2] and idx[1] < idx[2] and idx[2] <= idx[3], where
                    if op["check"] == "odd":
                        assert idx[2] <= idx[4], where
                continue
            if tree is None:
                tree = o.Tree(pts)
            q = np.array(op["point"], dtype=np.float64)
            if kind == "query":
                idx, dist = tree.query(q, op["k"])
                bidx, bdist = o.brute_knn(pts, q, op["k"])
                if "expect_idx" in op:
                    assert list(idx) == op["expect_idx"], where
                    assert list(bidx[0]) == op["expect_idx"], where
                if "expect_dist" in op:
                    assert len(dist) == len(op["expect_dist"]), where
                    for a, b in zip(dist, op["expect_dist"]):
                        assert _close(a, b, op["tol"]), where
                assert np.array_equal(dist, bdist[0]), where
            elif kind == "query_nearest":
                i, d = tree.query_nearest(q)
                bidx, bdist = o.brute_knn(pts, q, 1)
                if "expect_idx" in op:
                    assert i == op["expect_idx"], where
                    assert int(bidx[0, 0]) == op["expect_idx"], where
                if "expect_dist" in op:
                    assert _close(d, op["expect_dist"], op["tol"]), where
                assert d == bdist[0, 0], where
            elif kind == "query_radius":
                got = sorted(int(x) for x in tree.query_radius(q, op["r"]))  # sort_unstable in the reference
                assert got == op["expect_sorted"], where
                assert list(o.brute_radius(pts, q, op["r"])) == op["expect_sorted"], where
            elif kind == "nearest_in_subtree":
                r = tree.nearest_in_subtree(q, op["root"], op["radius"])
                assert (r is None) == op["expect_none"], where
            else:
                raise AssertionError(f"unknown op {kind}")
    assert n_ops >= 30


def test_appendix_b_exact_bits(oracle_mod):
    """SURVEY.md Appendix B hex values, recomputed here in pure Python floats
    (IEEE double, unfused) as an independent restatement of src/distance.rs:26-35."""
    o = oracle_mod

    def fold(a, b):
        s = 0.0
        for x, y in zip(a, b):
            d = x - y
            s += d * d
        return math.sqrt(s)

    cases = [([3.0, 3.0], [1.0, 2.0], "0x1.1e3779b97f4a8p+1"),
             ([3.0, 3.0], [1.0, 1.0], "0x1.6a09e667f3bcdp+1"),
             ([0.0, 0.0], [1.0, 1.1], "0x1.7c9244a4fb68cp+0"),
             ([0.0, 0.0], [9.0, 9.0], "0x1.974b2334f2346p+3"),
             ([1.1, 1.2], [1.0, 1.1], "0x1.21a1851ff6309p-3"),
             ([1.0, 2.0], [1.0, 2.1], "0x1.99999999999a0p-4"),
             ([0.95, 1.96], [1.0, 2.0], "0x1.06459fbeb847fp-4")]
    for a, b, hx in cases:
        want = float.fromhex(hx)
        assert fold(a, b) == want
        assert o.euclidean(np.array(a), np.array(b)) == want


def test_property_tree_equals_naive(oracle_mod, kats):
    """The reference's property test (src/ball_tree.rs:742-765), seeded, plus a
    stronger form: distances bit-identical, indices identical where unique."""
    o = oracle_mod
    pt = kats["property_test"]
    rng = np.random.default_rng(20261003)
    for rep in range(20):
        pts = rng.random((pt["n"], pt["dim"]))
        tree = o.Tree(pts)
        for _ in range(pt["queries"]):
            q = rng.random(pt["dim"])
            idx, dist = tree.query(q, pt["k"])
            bidx, bdist = o.brute_knn(pts, q, pt["k"])
            assert np.array_equal(dist, bdist[0])
            assert np.array_equal(idx, bidx[0])  # continuous data: no ties
            i, d = tree.query_nearest(q)
            assert (i, d) == (int(bidx[0, 0]), bdist[0, 0])
            r = float(bdist[0, -1]) * 1.0000001
            assert sorted(tree.query_radius(q, r).tolist()) == o.brute_radius(pts, q, r).tolist()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,dim,k", [(1, 3, 2), (2, 1, 2), (7, 2, 7), (64, 10, 5), (128, 10, 5),
                                     (1000, 3, 2), (513, 16, 10), (300, 128, 10)])
def test_tree_vs_brute_shapes(oracle_mod, dtype, n, dim, k):
    """Faithful tree == canonical brute force on the bench shapes
    (benches/ball_tree.rs:8-62; BASELINE.json configs[0]) and beyond, f32 and f64."""
    o = oracle_mod
    from conftest import uniform
    pts = uniform((n, dim), 0x5EED0001 + n, dtype)
    qs = uniform((16, dim), 0x5EED0002 + n, dtype)
    tree = o.Tree(pts)
    bidx, bdist = o.brute_knn(pts, qs, k)
    tidx, tdist = tree.query_batch(qs, k, nthreads=2)
    assert np.array_equal(tdist, bdist)
    assert np.array_equal(tidx, bidx)
    # self-queries as in the reference bench (benches/ball_tree.rs:53-59)
    sidx, sdist = tree.query_batch(pts[:8], k)
    assert np.all(sdist[:, 0] == 0)
    assert np.array_equal(sidx[:, 0], np.arange(min(8, n)))


def test_ties_and_identical_points(oracle_mod):
    """8 identical points (src/ball_tree.rs:718-740): distances pinned, index order unpinned
    -> compare tree vs brute as a multiset of distances and a valid index set."""
    o = oracle_mod
    pts = np.ones((8, 2))
    tree = o.Tree(pts)
    idx, dist = tree.query(np.array([1.0, 2.0]), 3)
    assert list(dist) == [1.0, 1.0, 1.0]
    assert len(set(idx.tolist())) == 3 and all(0 <= i < 8 for i in idx)
    bidx, bdist = o.brute_knn(pts, np.array([1.0, 2.0]), 3)
    assert list(bidx[0]) == [0, 1, 2]  # canonical: ascending index inside a tie group


def test_nan_and_k_edge_cases(oracle_mod):
    """CHANGELOG.md:113-116: NaN coordinates do not panic and sort last; k=0 -> empty; k>n -> n."""
    o = oracle_mod
    pts = np.array([[0.0, 0.0], [np.nan, 1.0], [1.0, 1.0], [2.0, 2.0]])
    q = np.array([0.1, 0.1])
    tree = o.Tree(pts)
    idx, dist = tree.query(q, 10)
    assert len(idx) == 4 and np.isnan(dist[-1]) and idx[-1] == 1
    bidx, bdist = o.brute_knn(pts, q, 10)
    assert list(bidx[0]) == [0, 2, 3, 1]
    assert list(idx) == [0, 2, 3, 1]
    i0, d0 = tree.query(q, 0)
    assert len(i0) == 0 and len(d0) == 0


def test_high_dim_walk_visits_everything(oracle_mod):
    """at d=128 the walk prunes nothing (justifies the brute-force GPU design)."""
    o = oracle_mod
    from conftest import uniform
    n = 4096
    pts = uniform((n, 128), 1, np.float32)
    qs = uniform((4, 128), 2, np.float32)
    tree = o.Tree(pts)
    tree.eval_counts(reset=True)
    for q in qs:
        tree.query(q, 10)
    cen, pnt = tree.eval_counts()
    assert cen / len(qs) > 1.5 * tree.num_nodes  # every node's centroid distance evaluated ~twice


def test_fill_uniform_is_24_bit(oracle_mod):
    x = oracle_mod.fill_uniform(4096, 0x5EED0001)
    assert x.dtype == np.float32 and x.min() >= 0 and x.max() < 1
    assert np.all((x * 16777216.0) == np.floor(x * 16777216.0))
    y = oracle_mod.fill_uniform(100, 0x5EED0001, first_ctr=10)
    assert np.array_equal(y, x[10:110])


def test_cosine_restatement(oracle_mod):
    """Cosine::distance (src/distance.rs:85-107) against an independent f64 evaluation, and its structure: three
    sequential sums, zip truncation for the dot product only, zero vector -> NaN, pairwise mirrored with zero diagonal."""
    o = oracle_mod
    rng = np.random.default_rng(5)
    for dim in (1, 2, 7, 128, 769):
        a, b = rng.standard_normal(dim), rng.standard_normal(dim)
        want = 1.0 - float(a @ b) / (np.sqrt(float(a @ a)) * np.sqrt(float(b @ b)))
        assert abs(float(o.cosine(a, b)) - want) < 1e-12
        a32, b32 = a.astype(np.float32), b.astype(np.float32)
        assert abs(float(o.cosine(a32, b32)) - want) < 1e-4
        # the f32 fold step by step
        dot = n1 = n2 = np.float32(0)
        for k in range(dim):
            dot = np.float32(dot + np.float32(a32[k] * b32[k]))
        for k in range(dim):
            n1 = np.float32(n1 + np.float32(a32[k] * a32[k]))
        for k in range(dim):
            n2 = np.float32(n2 + np.float32(b32[k] * b32[k]))
        ref = np.float32(np.float32(1) - np.float32(dot / np.float32(np.sqrt(n1) * np.sqrt(n2))))
        assert o.cosine(a32, b32).tobytes() == ref.tobytes()
    assert np.isnan(o.cosine(np.zeros(3), np.ones(3)))
    a, b = rng.standard_normal(9), rng.standard_normal(5)
    want = 1.0 - float(a[:5] @ b) / (np.sqrt(float(a @ a)) * np.sqrt(float(b @ b)))
    assert abs(float(o.cosine(a, b)) - want) < 1e-12
    x = rng.standard_normal((6, 4))
    pw = o.pairwise_cosine(x)
    assert np.array_equal(pw, pw.T) and np.all(np.diag(pw) == 0)
    assert pw[1, 4] == o.cosine(x[1], x[4])
    assert np.array_equal(o.pairwise_cosine(x[:1]), np.zeros((1, 1)))
