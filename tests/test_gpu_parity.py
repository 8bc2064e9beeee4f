"""Parity tests proper (need an MI355X): the HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs, against the reference's golden
vectors, and -- at BASELINE sizes -- through size-independent properties.

Bar: indices and distances BIT-EXACT (distances compared as raw bytes); inside
groups of exactly equal distances the reference's index order is unspecified
(SURVEY.md A.3), this library returns ascending index = the oracle's canonical
brute force, so the comparison is still exact equality.
"""
import numpy as np
import pytest

from conftest import uniform

pytestmark = pytest.mark.gpu


def _eq_bits(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def _same_dist(a, b):
    """bit-exact, with NaN == NaN regardless of payload"""
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    na, nb = np.isnan(a), np.isnan(b)
    if not np.array_equal(na, nb):
        return False
    return a[~na].tobytes() == b[~nb].tobytes()


def _check_knn(pn, oracle_mod, pts, qs, k, engine="exact", opts=None):
    tree = pn.BallTree.euclidean(pts)
    tree.set_engine(engine)
    for o, v in (opts or {}).items():
        tree.set_option(o, v)
    idx, dist = tree.query_batch(qs, k)
    oidx, odist = oracle_mod.brute_knn(pts, qs, k)
    assert _same_dist(dist, odist), f"distances differ (engine={engine})"
    assert np.array_equal(idx, oidx), f"indices differ (engine={engine})"
    return tree


# --------------------------------------------------------------- golden vectors
def test_reference_golden_vectors_on_gpu(pn, kats):
    """Every k-NN / nearest / radius / pairwise vector of the reference's tests, f64, on the GPU."""
    n_ops = 0
    for v in kats["vectors"]:
        if "shape" in v or v.get("fortran") or not v["points"]:
            continue
        pts = np.array(v["points"], dtype=np.float64)
        tree = None
        for op in v["ops"]:
            kind = op["op"]
            where = f'{v["id"]} {kind}'
            if kind == "pairwise":
                got = pn.distance.pairwise(pts, pn.distance.Euclidean())
                assert np.array_equal(got, np.array(op["expect"])), where
                n_ops += 1
                continue
            if kind not in ("query", "query_nearest", "query_radius"):
                continue
            if tree is None:
                tree = pn.BallTree.euclidean(pts)
            q = np.array(op["point"], dtype=np.float64)
            n_ops += 1
            if kind == "query":
                idx, dist = tree.query(q, op["k"])
                if "expect_idx" in op:
                    assert list(idx) == op["expect_idx"], where
                if "expect_dist" in op:
                    assert len(dist) == len(op["expect_dist"]), where
                    for a, b in zip(dist, op["expect_dist"]):
                        assert abs(a - b) <= op["tol"], where
            elif kind == "query_nearest":
                i, d = tree.query_nearest(q)
                if "expect_idx" in op:
                    assert i == op["expect_idx"], where
                if "expect_dist" in op:
                    assert abs(d - op["expect_dist"]) <= op["tol"], where
            else:
                got = sorted(int(x) for x in tree.query_radius(q, op["r"]))
                assert got == op["expect_sorted"], where
    assert n_ops >= 20


def test_vantage_point_tree_facade(pn, kats):
    """euclidian (src/vantage_point_tree.rs:220-233): nearest of [0.95, 1.96] is row 0."""
    v = [x for x in kats["vectors"] if x["id"] == "G13"][0]
    vp = pn.VantagePointTree.euclidean(np.array(v["points"]))
    assert vp.query_nearest(np.array(v["ops"][0]["point"]))[0] == v["ops"][0]["expect_idx"]
    with pytest.raises(pn.ArrayError.Empty):
        pn.VantagePointTree.euclidean(np.zeros((0, 2)))


def test_reference_property_test_on_gpu(pn, oracle_mod, kats):
    """ball_tree_query (src/ball_tree.rs:742-765): 40x3 f64, 10 queries, k=5 == naive scan."""
    pt = kats["property_test"]
    rng = np.random.default_rng(7)
    pts = rng.random((pt["n"], pt["dim"]))
    qs = rng.random((pt["queries"], pt["dim"]))
    _check_knn(pn, oracle_mod, pts, qs, pt["k"])


# ------------------------------------------------------------- exact engine
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,dim,nq,k", [
    (1, 1, 3, 1), (3, 2, 5, 2), (64, 10, 64, 5), (128, 10, 16, 5), (1000, 3, 50, 2),   # bench shapes
    (257, 7, 33, 10), (4097, 33, 70, 17), (5000, 128, 130, 10), (3000, 96, 9, 100),
    (2000, 131, 20, 64), (900, 16, 11, 65), (1500, 8, 7, 200), (70, 5, 4, 1000),
])
def test_exact_engine_vs_oracle(pn, oracle_mod, dtype, n, dim, nq, k):
    pts = uniform((n, dim), 0x5EED0001 + n + dim, dtype)
    qs = uniform((nq, dim), 0x5EED0002 + n + dim, dtype)
    _check_knn(pn, oracle_mod, pts, qs, k)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_k_beyond_one_buffer_is_served_in_rounds(pn, oracle_mod, dtype):
    """k > 960 (one candidate buffer): rounds that resume after the last (distance, index); includes
    k == n (a full sort of the corpus per query) and heavy ties across a round boundary."""
    pts = uniform((5000, 6), 123, dtype)
    qs = uniform((5, 6), 124, dtype)
    _check_knn(pn, oracle_mod, pts, qs, 2000)
    _check_knn(pn, oracle_mod, pts[:3000], qs[:2], 3000)
    ties = np.repeat(uniform((40, 4), 9, dtype), 60, axis=0)  # 40 distinct rows, 60 copies each
    _check_knn(pn, oracle_mod, ties, qs[:3, :4], 1500)


def test_exact_engine_signed_wide_range(pn, oracle_mod):
    rng = np.random.default_rng(3)
    pts = (rng.standard_normal((3000, 20)) * np.exp(rng.uniform(-6, 6, (3000, 1)))).astype(np.float32)
    qs = (rng.standard_normal((40, 20)) * 3).astype(np.float32)
    _check_knn(pn, oracle_mod, pts, qs, 12)


@pytest.mark.parametrize("segments", [1, 2, 7])
def test_segment_count_does_not_change_results(pn, oracle_mod, segments):
    from petal_neighbors_amd import _lib
    pts = uniform((9000, 24), 5, np.float32)
    qs = uniform((77, 24), 6, np.float32)
    _check_knn(pn, oracle_mod, pts, qs, 10, "exact", {_lib.PN_OPT_SEGMENTS: segments})


def test_ties_identical_points_and_grids(pn, oracle_mod):
    """ball_tree_identical_points (src/ball_tree.rs:718-740) and heavy exact ties."""
    pts = np.ones((8, 2))
    tree = pn.BallTree.new(pts, pn.distance.Euclidean())
    i, d = tree.query_nearest(np.array([1.0, 2.0]))
    assert d == 1.0 and i == 0
    i, d = tree.query_nearest(np.array([1.0, 1.0]))
    assert d == 0.0 and i == 0
    idx, dist = tree.query(np.array([1.0, 2.0]), 3)
    assert list(idx) == [0, 1, 2] and list(dist) == [1.0, 1.0, 1.0]
    # integer grid: thousands of exactly tied distances, k straddling tie groups
    g = np.stack(np.meshgrid(np.arange(20.0), np.arange(20.0), np.arange(10.0), indexing="ij"), -1).reshape(-1, 3)
    g = np.concatenate([g, g[:700]]).astype(np.float32)  # duplicates too
    qs = np.array([[3.0, 3.0, 3.0], [0.5, 0.5, 0.5], [19.0, 19.0, 9.0], [7.25, 3.0, 1.0]], dtype=np.float32)
    for k in (1, 7, 27, 100, 300):
        _check_knn(pn, oracle_mod, g, qs, k)
    same = np.full((5000, 6), 0.25, dtype=np.float32)
    _check_knn(pn, oracle_mod, same, same[:3] + 1, 70)


def test_nan_inf_and_k_edges(pn, oracle_mod):
    """CHANGELOG.md:113-116: NaN never panics and sorts last; k=0 empty; k>n returns n."""
    pts = uniform((300, 5), 9, np.float32)
    pts[7, 2] = np.nan
    pts[100, 0] = np.inf
    pts[200, 4] = -np.inf
    qs = uniform((6, 5), 10, np.float32)
    qs[3, 1] = np.nan
    tree = pn.BallTree.euclidean(pts)
    idx, dist = tree.query_batch(qs, 300)
    oidx, odist = oracle_mod.brute_knn(pts, qs, 300)
    assert _same_dist(dist, odist) and np.array_equal(idx, oidx)
    assert np.isnan(dist[0, -1]) and idx[0, -1] == 7 and np.isinf(dist[0, -2])
    i0, d0 = tree.query(qs[0], 0)
    assert len(i0) == 0 and len(d0) == 0
    i5, d5 = tree.query(qs[0], 5000)
    assert len(i5) == 300
    for k in (1, 10):
        idx, dist = tree.query_batch(qs, k)
        oidx, odist = oracle_mod.brute_knn(pts, qs, k)
        assert _same_dist(dist, odist) and np.array_equal(idx, oidx)


def test_zip_truncation_and_strided_input(pn, oracle_mod):
    """A query shorter/longer than D is silently truncated (src/distance.rs:27-28)."""
    pts = uniform((500, 12), 21, np.float64)
    tree = pn.BallTree.euclidean(pts)
    q = uniform((12,), 22, np.float64)
    idx, dist = tree.query(q[:5], 4)
    oidx, odist = oracle_mod.brute_knn(pts[:, :5], q[:5], 4)
    assert _eq_bits(dist, odist[0]) and np.array_equal(idx, oidx[0])
    idx, dist = tree.query(np.concatenate([q, [9.0, 9.0]]), 4)
    oidx, odist = oracle_mod.brute_knn(pts, q, 4)
    assert _eq_bits(dist, odist[0]) and np.array_equal(idx, oidx[0])
    # row stride > ncols is fine (only the inner stride is checked, src/ball_tree.rs:47)
    wide = uniform((400, 16), 23, np.float32)
    view = wide[:, :10]
    t2 = pn.BallTree.euclidean(view)
    idx, dist = t2.query_batch(wide[:9, :10], 3)
    oidx, odist = oracle_mod.brute_knn(np.ascontiguousarray(view), np.ascontiguousarray(wide[:9, :10]), 3)
    assert _eq_bits(dist, odist) and np.array_equal(idx, oidx)


def test_sqrt_is_correctly_rounded_on_device(pn):
    """1-D corpus: distance = sqrt(x^2 rounded) -- compare the device sqrt with IEEE host sqrt."""
    rng = np.random.default_rng(5)
    x = np.exp(rng.uniform(-40, 40, 20000)).astype(np.float32)
    x[:4] = [0.0, 1e-45, 3.4e38, 1.1754944e-38]
    with np.errstate(over="ignore", under="ignore"):
        want = np.sqrt((x * x).astype(np.float32))
    # pairwise against the origin: one row of the matrix holds sqrt(x_i^2) for every i
    for lo in range(0, len(x), 1000):
        blk = np.concatenate([np.zeros(1, dtype=np.float32), x[lo:lo + 1000]]).reshape(-1, 1)
        got = pn.distance.pairwise(blk)[0, 1:]
        assert _same_dist(got, want[lo:lo + 1000])
    # two-term sums exercise odd mantissas
    y = rng.uniform(0, 1, (2000, 2)).astype(np.float32)
    blk = np.concatenate([np.zeros((1, 2), dtype=np.float32), y])
    got = pn.distance.pairwise(blk)[0, 1:]
    s = (y[:, 0] * y[:, 0]).astype(np.float32) + (y[:, 1] * y[:, 1]).astype(np.float32)
    assert _eq_bits(got, np.sqrt(s.astype(np.float32)))


# ------------------------------------------------------------- MFMA filter engine
@pytest.mark.parametrize("n,dim,nq,k", [
    (5000, 128, 300, 10), (20000, 128, 130, 10), (7001, 96, 257, 10), (4100, 64, 64, 1),
    (9000, 33, 100, 5), (3000, 8, 50, 10), (2500, 3, 40, 2), (6000, 1, 30, 7),
    (12000, 128, 40, 100), (5000, 100, 33, 64), (70000, 16, 200, 10),
])
def test_mfma_engine_vs_oracle(pn, oracle_mod, n, dim, nq, k):
    """MFMA filter + exact re-rank + verification == canonical brute force, bit for bit.
    Queries mix corpus rows (distance 0: worst cancellation for the GEMM expansion) and fresh draws."""
    pts = uniform((n, dim), 0xA11CE + n + dim, np.float32)
    qs = np.concatenate([pts[: nq // 3], uniform((nq - nq // 3, dim), 0xB0B + n, np.float32)])
    tree = _check_knn(pn, oracle_mod, pts, qs, k, "mfma")
    st = tree.stats()
    assert st["hot_launches"] == 0 and st["queries"] == nq
    assert st["candidates"] >= nq * min(k, n)


@pytest.mark.parametrize("n,dim,nq,k", [(20000, 768, 200, 10), (9000, 200, 150, 10), (5000, 129, 70, 5),
                                        (3000, 1024, 40, 20), (70000, 256, 300, 1)])
def test_mfma_engine_wide_rows(pn, oracle_mod, n, dim, nq, k):
    """D > 128 (BASELINE config 4 is D = 768): slab-accumulating MFMA filter kernel."""
    pts = uniform((n, dim), 0xD1CE + n + dim, np.float32)
    qs = np.concatenate([pts[: nq // 3], uniform((nq - nq // 3, dim), 0xFACE + n, np.float32)])
    tree = _check_knn(pn, oracle_mod, pts, qs, k, "mfma")
    assert tree.stats()["candidates"] >= nq * k
    _check_knn(pn, oracle_mod, pts[:4000], qs[:20], k, "auto")


def test_mfma_engine_gaussian_clusters_and_scales(pn, oracle_mod):
    rng = np.random.default_rng(11)
    centers = rng.standard_normal((20, 48)) * 10
    pts = (centers[rng.integers(0, 20, 30000)] + rng.standard_normal((30000, 48)) * 0.05).astype(np.float32)
    qs = np.concatenate([pts[:64], (centers + 0.01).astype(np.float32)])
    _check_knn(pn, oracle_mod, pts, qs, 10, "mfma")
    big = (uniform((8000, 64), 77, np.float32) * 1e4 - 5e3).astype(np.float32)
    _check_knn(pn, oracle_mod, big, big[:50] + np.float32(0.25), 10, "mfma")
    tiny = (uniform((8000, 64), 78, np.float32) * 1e-20).astype(np.float32)
    _check_knn(pn, oracle_mod, tiny, tiny[:50], 10, "mfma")


def test_mfma_engine_falls_back_on_unprovable_queries(pn, oracle_mod):
    """Heavy exact ties defeat the filter's proof (more tied rows than candidate slots): those
    queries must be flagged and re-run on the exact engine -- results stay bit-exact."""
    g = np.stack(np.meshgrid(np.arange(16.0), np.arange(16.0), np.arange(16.0), np.arange(4.0), indexing="ij"),
                 -1).reshape(-1, 4).astype(np.float32)
    g = np.concatenate([g, g, g])  # every row three times
    qs = np.array([[3, 3, 3, 1], [0.5, 0.5, 0.5, 0.5], [15, 15, 15, 3], [7.25, 3, 1, 2]], dtype=np.float32)
    from petal_neighbors_amd import _lib
    tree = _check_knn(pn, oracle_mod, g, qs, 10, "mfma", {_lib.PN_OPT_MFMA_STRUCTURE: 2, _lib.PN_OPT_FILTER_SLOTS: 10})
    assert tree.stats()["fallback_queries"] >= 1
    _check_knn(pn, oracle_mod, g, qs, 10, "mfma")
    same = np.full((6000, 32), 0.25, dtype=np.float32)
    tree = _check_knn(pn, oracle_mod, same, same[:5] + 1, 20, "mfma")
    assert tree.stats()["fallback_queries"] == 5


def test_mfma_engine_nan_inf_queries_and_nonfinite_corpus(pn, oracle_mod):
    pts = uniform((5000, 16), 91, np.float32)
    qs = uniform((8, 16), 92, np.float32)
    qs[2, 3] = np.nan
    qs[5, 0] = np.inf
    qs[6, :] = 3e19  # squared norm overflows f32
    tree = _check_knn(pn, oracle_mod, pts, qs, 10, "mfma")
    assert tree.stats()["fallback_queries"] >= 3
    bad = pts.copy()
    bad[17, 2] = np.nan
    t2 = pn.BallTree.euclidean(bad)
    assert not t2.mfma_eligible  # non-finite norm: the index is served by the exact engine
    with pytest.raises(pn.PetalError):
        t2.set_engine("mfma")
    idx, dist = t2.query_batch(qs[:2], 10)
    oidx, odist = oracle_mod.brute_knn(bad, qs[:2], 10)
    assert _same_dist(dist, odist) and np.array_equal(idx, oidx)


@pytest.mark.parametrize("slots", [10, 16, 40])
def test_mfma_filter_slot_count_does_not_change_results(pn, oracle_mod, slots):
    from petal_neighbors_amd import _lib
    pts = uniform((30000, 128), 5, np.float32)
    qs = uniform((70, 128), 6, np.float32)
    _check_knn(pn, oracle_mod, pts, qs, 10, "mfma", {_lib.PN_OPT_FILTER_SLOTS: slots, _lib.PN_OPT_SEGMENTS: 3})


@pytest.mark.parametrize("structure", [2, 3])
@pytest.mark.parametrize("n,dim,nq,k", [(40000, 128, 300, 10), (9000, 96, 130, 20), (2000, 16, 700, 3), (300, 8, 5, 1)])
def test_mfma_both_kernel_structures(pn, oracle_mod, structure, n, dim, nq, k):
    """structure 1 = (query tile x segment) grid; 2 / 3 = persistent balanced partition with LDS / HBM
    candidate buffers (k' <= 30)."""
    from petal_neighbors_amd import _lib
    pts = uniform((n, dim), 0xC0FFEE + n, np.float32)
    qs = np.concatenate([pts[: nq // 4], uniform((nq - nq // 4, dim), 0xF00D + n, np.float32)])
    _check_knn(pn, oracle_mod, pts, qs, k, "mfma", {_lib.PN_OPT_MFMA_STRUCTURE: structure})


def test_auto_engine_picks_mfma_and_matches(pn, oracle_mod):
    pts = uniform((50000, 128), 15, np.float32)
    qs = uniform((500, 128), 16, np.float32)
    tree = _check_knn(pn, oracle_mod, pts, qs, 10, "auto")
    assert tree.mfma_eligible and tree.stats()["candidates"] > 0


# --------------------------------------------------------------- radius / pairwise
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_query_radius_vs_oracle(pn, oracle_mod, dtype):
    """64x10 r=0.2 is the reference bench shape (benches/ball_tree.rs:22-41)."""
    for n, dim, r in ((64, 10, 0.2), (64, 10, 0.9), (3000, 3, 0.15), (5000, 16, 1.2), (700, 128, 4.2)):
        pts = uniform((n, dim), 31 + n, dtype)
        qs = np.concatenate([pts[:20], uniform((13, dim), 32 + n, dtype)])
        tree = pn.BallTree.euclidean(pts)
        off, idx = tree.query_radius_batch(qs, r)
        assert off[0] == 0 and off[-1] == len(idx)
        tot = 0
        for a in range(len(qs)):
            want = oracle_mod.brute_radius(pts, qs[a], dtype(r))
            got = idx[int(off[a]):int(off[a + 1])]
            assert np.array_equal(got, want), (n, dim, r, a)
            tot += len(want)
        assert tot == len(idx)
        one = tree.query_radius(qs[0], r)
        assert np.array_equal(one, oracle_mod.brute_radius(pts, qs[0], dtype(r)))
    # boundary: a point at distance exactly r is excluded (strict '<', src/ball_tree.rs:277)
    line = np.array([[0.0], [2.0], [3.0], [4.0]], dtype=dtype)
    t = pn.BallTree.euclidean(line)
    assert list(t.query_radius(np.array([3.0], dtype=dtype), 1.0)) == [2]


@pytest.mark.parametrize("engine", ["bf16", "mfma", "auto", "exact"])
def test_query_radius_mfma_filter(pn, oracle_mod, engine):
    """query_radius through the bf16 / f32 MFMA filters: sparse results stay on them, dense results overflow the
    survivor lists and are re-run exactly; boundary radii (r == an exact distance) keep the strict '<'."""
    for n, dim in ((20000, 128), (9000, 96), (6000, 16), (5000, 3)):
        pts = uniform((n, dim), 51 + n, np.float32)
        qs = np.concatenate([pts[:10], uniform((23, dim), 52 + n, np.float32)])
        tree = pn.BallTree.euclidean(pts)
        tree.set_engine(engine)
        _, d = oracle_mod.brute_knn(pts, qs, 40)
        radii = [float(np.median(d[:, 3])), float(d[12, 5]), float(d[:, 30].max()) * 1.01, 1e-6, 0.0, -1.0,
                 float("inf"), float("nan")]
        for r in radii:
            off, idx = tree.query_radius_batch(qs, r)
            assert off[0] == 0 and off[-1] == len(idx)
            for a in range(len(qs)):
                want = oracle_mod.brute_radius(pts, qs[a], np.float32(r))
                assert np.array_equal(idx[int(off[a]):int(off[a + 1])], want), (n, dim, r, a, engine)
    # exact boundary: r equal to a stored distance must exclude that row
    pts = uniform((8000, 64), 77, np.float32)
    q = uniform((64,), 78, np.float32)
    tree = pn.BallTree.euclidean(pts).set_engine(engine)
    i5, d5 = tree.query(q, 5)
    got = tree.query_radius(q, float(d5[4]))
    assert list(got) == sorted(int(i) for i in i5[:4])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_pairwise_vs_oracle(pn, oracle_mod, dtype):
    for n, dim in ((1, 3), (2, 2), (65, 7), (300, 128), (130, 131)):
        x = uniform((n, dim), 41 + n, dtype)
        got = pn.distance.pairwise(x, pn.distance.Euclidean())
        want = oracle_mod.pairwise(x)
        assert _eq_bits(got, want), (n, dim)
        assert np.array_equal(got, got.T) and np.all(np.diag(got) == 0)


def _eq_bits_or_nan(a, b):
    """bit-identical except that a NaN only has to be a NaN (0/0 has a different sign bit on x86 and on the GPU)"""
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and a[~na].tobytes() == b[~nb].tobytes()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cosine_metric_and_pairwise_vs_oracle(pn, oracle_mod, kats, dtype):
    """distance::Cosine (src/distance.rs:76-122): pair metric on the host, pairwise on the GPU, both the oracle's
    bit for bit; the reference's own cosine test vectors; zero vectors give NaN as in the reference (0/0)."""
    m = pn.distance.Cosine()
    n_vec = 0
    for v in kats["vectors"]:
        for op in v["ops"]:
            if op["op"] != "cosine":
                continue
            dt = np.float32 if op.get("dtype") == "f32" else np.float64
            got = m.distance(np.array(op["a"], dtype=dt), np.array(op["b"], dtype=dt))
            assert abs(float(got) - op["expect"]) <= op["tol"], op
            assert m.rdistance(np.array(op["a"], dtype=dt), np.array(op["b"], dtype=dt)) == got
            n_vec += 1
    assert n_vec >= 6
    assert m.rdistance_to_distance(dtype(0.25)) == dtype(0.25) and m.distance_to_rdistance(dtype(0.25)) == dtype(0.25)
    assert m == pn.distance.Cosine() and m != pn.distance.Euclidean()
    for n, dim in ((1, 3), (2, 2), (65, 7), (300, 128), (130, 131), (40, 768)):
        x = (uniform((n, dim), 61 + n, dtype) - dtype(0.3)).astype(dtype)
        if n > 10:
            x[5] = 0  # a zero vector: 0/0 -> NaN in its row and column (not on the diagonal)
            x[7] = x[3] * dtype(2)  # parallel vectors
        got = pn.distance.pairwise(x, m)
        want = oracle_mod.pairwise_cosine(x)
        assert _eq_bits_or_nan(got, want), (n, dim)
        assert np.all(np.diag(got) == 0)
        for i, j in ((0, n - 1), (n // 2, n // 3), (3 % n, 7 % n)):
            if i != j:
                assert _eq_bits_or_nan(np.array([m.distance(x[i], x[j])]), np.array([want[i, j]])), (n, dim, i, j)
    # unequal lengths: the dot product zips (shorter length), each norm runs over its own vector
    a, b = uniform((9,), 1, dtype), uniform((5,), 2, dtype)
    assert m.distance(a, b).tobytes() == oracle_mod.cosine(a, b).tobytes()
    with pytest.raises(NotImplementedError):
        pn.BallTree.new(uniform((10, 3), 3, dtype), object())


def _cosine_brute(oracle_mod, pts, qs, k):
    """k smallest (Cosine::distance, index) per query from the oracle's pairwise(x, &Cosine) over [queries; points]"""
    nq = len(qs)
    d = oracle_mod.pairwise_cosine(np.vstack([qs, pts]))[:nq, nq:]
    idx = np.empty((nq, min(k, len(pts))), dtype=np.uint64)
    dist = np.empty(idx.shape, dtype=pts.dtype)
    for a in range(nq):
        nan = np.isnan(d[a])  # ordered-float: NaN greatest; cosine distances may be a few ulp below zero
        order = np.lexsort((np.arange(len(pts)), np.where(nan, np.inf, d[a]), nan))[: idx.shape[1]]
        idx[a], dist[a] = order, d[a][order]
    return idx, dist


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,dim,nq,k", [(3000, 24, 40, 7), (500, 3, 30, 500), (70, 130, 9, 5), (5000, 128, 64, 10)])
def test_ball_tree_under_cosine_is_an_exact_scan(pn, oracle_mod, dtype, n, dim, nq, k):
    """BallTree::new(points, Cosine): k smallest Cosine::distance in the reference's arithmetic (src/distance.rs:85-107),
    against the oracle's pairwise(x, &Cosine) bit for bit -- k-NN, nearest, radius (the deviation from the reference's
    pruned walk is documented at pn_index_create_cosine_*)."""
    pts = uniform((n, dim), 700 + n, dtype) - dtype(0.3)
    pts[3] = 0  # a zero row: its distance to anything is NaN and sorts last
    qs = np.concatenate([pts[10:13] * dtype(2.0), uniform((nq - 3, dim), 701 + n, dtype) - dtype(0.3)])
    tree = pn.BallTree.new(pts, pn.distance.Cosine())
    assert tree.metric == pn.distance.Cosine() and not tree.bf16_eligible and not tree.mfma_eligible
    want_i, want_d = _cosine_brute(oracle_mod, pts, qs, k)
    gi, gd = tree.query_batch(qs, k)
    assert _same_dist(gd, want_d), "cosine distances differ"
    fin = ~np.isnan(want_d)
    assert np.array_equal(gi[fin], want_i[fin])
    i0, d0 = tree.query_nearest(qs[5])
    assert i0 == int(want_i[5, 0]) and d0.tobytes() == want_d[5, 0].tobytes()
    dall = oracle_mod.pairwise_cosine(np.vstack([qs, pts]))[:nq, nq:]
    r = dtype(np.sort(dall[4])[min(6, n - 1)])
    off, ids = tree.query_radius_batch(qs, r)
    for a in range(nq):
        assert np.array_equal(ids[int(off[a]):int(off[a + 1])], np.nonzero(dall[a] < r)[0].astype(np.uint64)), a
    with pytest.raises(pn.PetalError):
        tree.set_engine("bf16")
    # zip truncation: a shorter query uses the rows' full norms and its own
    if dim > 4:
        qi, qd = tree.query(qs[7][: dim - 2], 3)
        m = pn.distance.Cosine()
        allv = np.array([m.distance(qs[7][: dim - 2], p) for p in pts])
        keep = ~np.isnan(allv)
        best = np.lexsort((np.arange(n)[keep], allv[keep]))[:3]
        assert np.array_equal(qi, np.arange(n)[keep][best].astype(np.uint64)) and _same_dist(qd, allv[keep][best])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_pairwise_symmetric_tiles_and_device_entry(pn, oracle_mod, dtype):
    """distance::pairwise at a size with many tiles (the upper triangle of 64 x 64 tiles is computed, each pair
    written twice), host entry and HBM-resident entry, against the oracle bit for bit."""
    import torch
    for n, dim in ((1000, 33), (257, 128), (65, 3)):
        x = uniform((n, dim), 900 + n, dtype)
        want = oracle_mod.pairwise(x)
        got = pn.distance.pairwise(x, pn.distance.Euclidean())
        assert got.tobytes() == want.tobytes(), (n, dim)
        xd = torch.from_numpy(x).to("cuda:0")
        gd = pn.distance.pairwise_device(xd)
        torch.cuda.synchronize()
        assert gd.cpu().numpy().tobytes() == want.tobytes(), (n, dim)
