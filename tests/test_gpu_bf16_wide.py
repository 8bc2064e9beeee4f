"""bf16 first tier on WIDE rows (128 < D <= 1024): the K-chunked kernel (bf16_wide_kernel in
petal-neighbors_amd/csrc/bf16_filter.hip).  Same contract as tests/test_gpu_bf16.py: the bound is a proven lower bound
(checked against f64), the matrix core's accumulation error over the longer chains stays far inside the allowance, and
the k-NN results are the oracle's bit for bit whatever the filter did."""
import numpy as np
import pytest

from conftest import uniform
from test_gpu_bf16 import CASES, _bounds, _check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["uniform", "centered", "offset1000", "mixed_scales", "sparse", "scaled1e6"])
@pytest.mark.parametrize("dim", [129, 200, 768, 1024, 1536, 4096])
def test_wide_lower_bound_inequality(pn, name, dim):
    n, nq = 1024, 64
    pts = CASES[name](n, dim, 31)
    qs = CASES[name](nq, dim, 32)
    tree = pn.BallTree.euclidean(pts)
    assert tree.bf16_eligible
    L, qn, mu = _bounds(pn, tree, qs, 600)
    p64, q64 = pts[:600].astype(np.float64), qs.astype(np.float64)
    qq = ((q64 - mu.astype(np.float64)) ** 2).sum(1)
    assert np.all(qn <= qq) and np.all(qn >= qq * (1 - 1e-11))
    d2 = ((q64[:, None, :] - p64[None, :, :]) ** 2).sum(2)
    assert np.all(np.isfinite(L))
    gap = d2 - (L.astype(np.float64) + qn[:, None])
    assert gap.min() >= 0.0, f"{name}/D={dim}: bound exceeds the squared distance by {-gap.min()}"
    if name in ("uniform", "centered", "offset1000"):
        assert gap.max() < 0.05 * d2.mean() + 1e-3


@pytest.mark.parametrize("name", ["centered", "offset1000", "mixed_scales", "uniform"])
@pytest.mark.parametrize("dim", [256, 768, 1024, 2048, 4096])
def test_wide_accumulation_error_is_far_inside_the_allowance(pn, name, dim):
    """chains of up to 257 MFMA steps (D = 4096; round 3: 65 at D = 1024): |delivered - exact sum of the bf16 x bf16 terms| <= g sum|terms|"""
    from test_bf16_bound_model import corpus_columns, query_columns, G
    n, nq = 512, 64
    pts = CASES[name](n, dim, 41)
    qs = CASES[name](nq, dim, 42)
    tree = pn.BallTree.euclidean(pts)
    L, _, mu = _bounds(pn, tree, qs, n)
    ph, pieces, bp, dp = corpus_columns(pts, mu)
    mq, aq, cq, _ = query_columns(qs, mu)
    exact = pieces.sum(1)[None, :] + mq @ ph.T - aq[:, None] * bp[None, :] - cq[:, None] * dp[None, :]
    mags = pieces.sum(1)[None, :] + np.abs(mq) @ np.abs(ph).T + aq[:, None] * bp[None, :] + cq[:, None] * dp[None, :]
    ratio = np.abs(L.astype(np.float64) - exact) / (G * mags)
    print(f"{name}/D={dim}: accumulation error / allowance: max {ratio.max():.4f}, mean {ratio.mean():.5f}")
    assert ratio.max() < 0.02, f"matrix-core accumulation error uses {ratio.max():.4f} of the allowance"  # measured: <= 0.0016


@pytest.mark.parametrize("n,dim,nq,k", [(20000, 256, 300, 10), (9000, 768, 257, 10), (6000, 1024, 64, 5),
                                          (12000, 200, 700, 1), (5000, 129, 1, 3), (30000, 384, 100, 100),
                                          (4097, 144, 513, 7), (300, 512, 40, 4), (6000, 1536, 130, 10),
                                          (3000, 3072, 40, 5), (2000, 4096, 20, 3), (5000, 1030, 33, 3)])
def test_wide_engine_parity(pn, oracle_mod, n, dim, nq, k):
    pts = uniform((n, dim), 300 + dim)
    qs = uniform((nq, dim), 400 + dim)
    _check(pn, oracle_mod, pts, qs, k)


def test_wide_is_the_auto_default_and_proves_uniform_data(pn, oracle_mod):
    pts = uniform((40000, 768), 51)
    qs = uniform((600, 768), 52)
    tree = pn.BallTree.euclidean(pts)
    idx, dist = tree.query_batch(qs, 10)
    oidx, odist = oracle_mod.brute_knn(pts, qs, 10)
    assert dist.tobytes() == odist.tobytes() and np.array_equal(idx, oidx)
    st = tree.stats()
    assert st["fallback_queries"] <= 6, st
    assert tree.info()["bf16_eligible"] if hasattr(tree, "info") else tree.bf16_eligible


def test_wide_ties_duplicates_and_small_slots(pn, oracle_mod):
    base = uniform((3000, 320), 61)
    pts = np.concatenate([base, base[:500], base[:40]]).astype(np.float32)  # exact duplicates: distance ties
    qs = np.concatenate([base[:64], uniform((64, 320), 62)]).astype(np.float32)
    _check(pn, oracle_mod, pts, qs, 6)
    from petal_neighbors_amd import _lib
    _check(pn, oracle_mod, pts, qs, 20, opts={_lib.PN_OPT_FILTER_SLOTS: 20})  # k' = k: proofs fail or not, results stay exact


def test_wide_clustered_order_and_offset(pn, oracle_mod):
    rng = np.random.default_rng(7)
    centres = rng.random((20, 300), dtype=np.float32) * 4
    pts = (centres[np.repeat(np.arange(20), 800)] + 0.05 * rng.standard_normal((16000, 300), dtype=np.float32)
           + np.float32(500.0)).astype(np.float32)  # sorted by cluster, far from the origin
    qs = (pts[rng.integers(0, 16000, 200)] + 0.01 * rng.standard_normal((200, 300), dtype=np.float32)).astype(np.float32)
    _check(pn, oracle_mod, pts, qs, 10)


def test_wide_nonfinite_queries(pn, oracle_mod):
    pts = uniform((8000, 256), 71)
    qs = uniform((70, 256), 72)
    qs[3, 5] = np.nan
    qs[9, 0] = np.inf
    qs[11, 200] = 3e38
    tree = pn.BallTree.euclidean(pts)
    tree.set_engine("bf16")
    idx, dist = tree.query_batch(qs, 4)
    oidx, odist = oracle_mod.brute_knn(pts, qs, 4)
    assert dist.tobytes() == odist.tobytes()
    ok = np.isfinite(odist).all(1)
    assert np.array_equal(idx[ok], oidx[ok])


def test_rows_beyond_4096_columns_have_no_bf16_tier(pn, oracle_mod):
    """(round 4: the tier's limit went from 1024 to 4096 columns -- 257 MFMA steps, a chain length the library's own
    self-test contracts; beyond it the f32 tiers answer)"""
    pts = uniform((1500, 4100), 81)
    tree = pn.BallTree.euclidean(pts)
    assert not tree.bf16_eligible
    qs = uniform((33, 4100), 82)
    idx, dist = tree.query_batch(qs, 3)
    oidx, odist = oracle_mod.brute_knn(pts, qs, 3)
    assert dist.tobytes() == odist.tobytes() and np.array_equal(idx, oidx)


@pytest.mark.parametrize("engine", ["bf16", "auto"])
def test_wide_query_radius(pn, oracle_mod, engine):
    """query_radius on wide rows through the bf16 filter's fixed-threshold mode: sparse results stay on it, dense
    results overflow the survivor lists and are re-run exactly; boundary radii keep the strict '<'."""
    for n, dim in ((12000, 256), (7000, 768), (5000, 130)):
        pts = uniform((n, dim), 91 + dim, np.float32)
        qs = np.concatenate([pts[:10], uniform((23, dim), 92 + dim, np.float32)])
        tree = pn.BallTree.euclidean(pts)
        tree.set_engine(engine)
        _, d = oracle_mod.brute_knn(pts, qs, 40)
        radii = [float(np.median(d[:, 3])), float(d[12, 5]), float(d[:, 30].max()) * 1.01, float(d.max()) * 1.2, 1e-6,
                 0.0, -1.0, float("inf"), float("nan")]
        for r in radii:
            off, idx = tree.query_radius_batch(qs, r)
            assert off[0] == 0 and off[-1] == len(idx)
            for a in range(len(qs)):
                want = oracle_mod.brute_radius(pts, qs[a], np.float32(r))
                assert np.array_equal(idx[int(off[a]):int(off[a + 1])], want), (n, dim, r, a, engine)
    pts = uniform((9000, 512), 97, np.float32)
    q = uniform((512,), 98, np.float32)
    tree = pn.BallTree.euclidean(pts).set_engine(engine)
    i5, d5 = tree.query(q, 5)
    got = tree.query_radius(q, float(d5[4]))
    assert list(got) == sorted(int(i) for i in i5[:4])


def test_wide_partition_edge_cases(pn, oracle_mod):
    """few row tiles (empty segments), one query tile against many row ranges, the segment cap option, large k'"""
    from petal_neighbors_amd import _lib
    # 3 row tiles, 700 queries (3 query tiles): runs of one tile, no scout
    _check(pn, oracle_mod, uniform((700, 160), 1), uniform((700, 160), 2), 3)
    # one query against 40 000 rows: 1 query tile x up to 32 row ranges
    _check(pn, oracle_mod, uniform((40000, 136), 3), uniform((1, 136), 4), 10)
    # segment cap: at most 2 segments per query (one workgroup per query tile)
    _check(pn, oracle_mod, uniform((30000, 192), 5), uniform((300, 192), 6), 5, opts={_lib.PN_OPT_SEGMENTS: 2})
    # k = 150 (k' beyond 128 slots: 256-slot buffers)
    tree, st = _check(pn, oracle_mod, uniform((25000, 144), 7), uniform((70, 144), 8), 150)
    assert st["queries"] == 70


@pytest.mark.parametrize("n,dim,nq,k", [(20000, 64, 66500, 5), (20000, 256, 70000, 5), (30000, 128, 140000, 3)])
def test_many_query_tiles_run_the_grid_in_rounds(pn, oracle_mod, n, dim, nq, k):
    """more query tiles than workgroup slots: the plan launches a grid of c workgroups per query tile that runs in
    rounds (narrow rows: bf16_grid_wgs, wide rows: bf16_plan_wide); a sample of the queries against the oracle"""
    pts = uniform((n, dim), 500 + dim)
    qs = uniform((nq, dim), 600 + dim)
    tree = pn.BallTree.euclidean(pts)
    idx, dist = tree.query_batch(qs, k)
    sel = np.unique(np.concatenate([np.arange(0, nq, 997), [nq - 1, nq - 2, 255, 256, 257]]))
    oidx, odist = oracle_mod.brute_knn(pts, qs[sel], k)
    assert dist[sel].tobytes() == odist.tobytes() and np.array_equal(idx[sel], oidx)
    assert np.all(np.diff(dist, axis=1) >= 0)
    st = tree.stats()
    assert st["fallback_queries"] <= nq // 200, st
    # radius through the same plan
    r = float(odist[0, -1]) * 1.0000001
    off, ids = tree.query_radius_batch(qs[:300], r)
    for a in (0, 1, 150, 299):
        assert np.array_equal(ids[int(off[a]):int(off[a + 1])], oracle_mod.brute_radius(pts, qs[a], np.float32(r)))
