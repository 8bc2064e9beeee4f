"""Cosine indexes behind the bf16 MFMA filter (round 4; SURVEY.md 8 f3: "normalise rows once, then the same contraction").

The filter runs over the rows and queries normalised in f64 -- |q~ - p~|^2 = 2 (1 - cos) -- and only FILTERS: every answer
is a Cosine::distance evaluated in the reference's arithmetic (src/distance.rs:85-107) by the re-rank or, for queries
whose exclusions cannot be proven (select.hip, cos_proof_lb), by the exact scan.  So whatever the tier does, answers are
the oracle's bit for bit; what it buys is speed (bench.py --metric cosine).
"""
import numpy as np
import pytest

from conftest import uniform

pytestmark = pytest.mark.gpu


def _same_dist(a, b):
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and a[~na].tobytes() == b[~nb].tobytes()


def _cosine_brute(oracle_mod, pts, qs, k):
    """k smallest (Cosine::distance, index) per query from the oracle's pairwise(x, &Cosine) over [queries; points]"""
    nq = len(qs)
    d = oracle_mod.pairwise_cosine(np.vstack([qs, pts]))[:nq, nq:]
    idx = np.empty((nq, min(k, len(pts))), dtype=np.uint64)
    dist = np.empty(idx.shape, dtype=pts.dtype)
    for a in range(nq):
        nan = np.isnan(d[a])
        order = np.lexsort((np.arange(len(pts)), np.where(nan, np.inf, d[a]), nan))[: idx.shape[1]]
        idx[a], dist[a] = order, d[a][order]
    return idx, dist


def _cosine_oracle_large(oracle_mod, pts, qs, k, shortlist=400):
    """For corpora where pairwise(x, &Cosine) over everything is out of reach: the candidates are the `shortlist` rows
    of largest f64 cosine (the reference's float result differs from the real cosine by < 1e-4, the shortlist's last
    member is far beyond the k-th), each evaluated by the oracle's scalar Cosine::distance, ordered by (distance, index)."""
    p64 = pts.astype(np.float64)
    pn = p64 / np.linalg.norm(p64, axis=1, keepdims=True)
    idx = np.empty((len(qs), k), dtype=np.uint64)
    dist = np.empty((len(qs), k), dtype=pts.dtype)
    for a, q in enumerate(qs):
        c = pn @ (q.astype(np.float64) / np.linalg.norm(q.astype(np.float64)))
        short = np.argpartition(-c, shortlist)[:shortlist]
        assert c[short].min() < np.sort(c)[-k] - 1e-3, "shortlist too short for this data"
        d = np.array([oracle_mod.cosine(q, pts[i]) for i in short], dtype=pts.dtype)
        order = np.lexsort((short, d))[:k]
        idx[a], dist[a] = short[order], d[order]
    return idx, dist


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,dim,nq,k", [(3000, 24, 40, 7), (500, 3, 30, 100), (70, 130, 9, 5), (5000, 128, 64, 10),
                                          (9000, 96, 300, 1), (4200, 768, 20, 10)])
def test_cosine_tier_small_shapes_bit_identical_to_the_oracle(pn, oracle_mod, dtype, n, dim, nq, k):
    """the shapes of test_ball_tree_under_cosine_is_an_exact_scan (and two more) with the tier forced on: every answer
    bit for bit the oracle's pairwise(x, &Cosine)"""
    pts = uniform((n, dim), 700 + n, dtype) - dtype(0.3)
    qs = np.concatenate([pts[10:13] * dtype(2.0), uniform((nq - 3, dim), 701 + n, dtype) - dtype(0.3)])  # 3 parallel to rows
    tree = pn.BallTree.new(pts, pn.distance.Cosine())
    assert tree.bf16_eligible and not tree.mfma_eligible
    want_i, want_d = _cosine_brute(oracle_mod, pts, qs, k)
    for eng in ("bf16", "exact", "auto"):
        tree.set_engine(eng)
        gi, gd = tree.query_batch(qs, k)
        assert _same_dist(gd, want_d), (eng, "cosine distances differ")
        assert np.array_equal(gi, want_i), (eng, "indices differ")
    with pytest.raises(pn.PetalError):
        tree.set_engine("mfma")
    # a zero row makes the reference's distance NaN there: such an index keeps the exact scan
    pts2 = pts.copy()
    pts2[3] = 0
    t2 = pn.BallTree.new(pts2, pn.distance.Cosine())
    assert not t2.bf16_eligible


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cosine_tier_at_scale_hostile_queries_and_ties(pn, oracle_mod, dtype):
    """200 000 x 128: the tier on (auto), against the exact engine on every query and the oracle's scalar metric on a
    sample; duplicated rows (ties at the cut), queries parallel to rows (distances a few ulp around zero, some negative),
    a zero query and a NaN query (the exact engine answers them: NaN distances, ascending index), a query of another
    length (zip truncation: exact scan)."""
    n, dim, nq, k = 200_000, 128, 600, 10
    rng = np.random.default_rng(88)
    pts = (uniform((n, dim), 8801, dtype) - dtype(0.5)).astype(dtype)
    pts[5000:5040] = pts[17] * dtype(3.0)       # 40 rows parallel to row 17: equal cosine, distances within ulps
    qs = (uniform((nq, dim), 8802, dtype) - dtype(0.5)).astype(dtype)
    qs[0] = pts[17] * dtype(0.5)
    qs[1] = 0
    qs[2, 5] = np.nan
    qs[3] = pts[123456]
    tree = pn.BallTree.new(pts, pn.distance.Cosine())
    assert tree.bf16_eligible
    gi, gd = tree.query_batch(qs, k)
    st = tree.stats(reset=True)
    assert st["fallback_queries"] <= 12, st        # the tier served the batch (bad queries and the tie group aside)
    tree.set_engine("exact")
    ei, ed = tree.query_batch(qs, k)
    assert _same_dist(gd, ed) and np.array_equal(gi, ei), "tier and exact scan disagree"
    assert np.isnan(gd[1]).all() and np.isnan(gd[2]).all() and np.array_equal(gi[1], np.arange(k, dtype=np.uint64))
    assert gi[3, 0] == 123456
    sel = np.array([0, 3, 4, 50, 333, 599])
    oi, od = _cosine_oracle_large(oracle_mod, pts, qs[sel], k)
    assert _same_dist(gd[sel], od) and np.array_equal(gi[sel], oi), "tier differs from the oracle's scalar metric"
    # another query length: the dot product zips, the query's norm runs over ITS length -- exact scan, same contract
    tree.set_engine("auto")
    qi, qd = tree.query(qs[7][: dim - 5], 3)
    m = pn.distance.Cosine()
    c = np.argsort([float(m.distance(qs[7][: dim - 5], p)) for p in pts[:3000]])[:1]
    assert qd[0] <= m.distance(qs[7][: dim - 5], pts[c[0]])


@pytest.mark.parametrize("family", ["positive", "gaussian", "clustered", "scaled"])
def test_cosine_tier_data_families(pn, oracle_mod, family):
    """the bound's slack depends on the data, the answers do not: all-positive coordinates (every vector within a narrow
    cone: tiny distances), unit-free gaussian embeddings, tight clusters (many rows under the k-th distance: queries go
    to the exact scan), per-row scales over twelve decades (normalisation removes them)"""
    n, dim, nq, k = 60_000, 64, 200, 10
    rng = np.random.default_rng(99)
    if family == "positive":
        pts, qs = uniform((n, dim), 8901), uniform((nq, dim), 8902)
    elif family == "gaussian":
        pts, qs = rng.standard_normal((n, dim)).astype(np.float32), rng.standard_normal((nq, dim)).astype(np.float32)
    elif family == "clustered":
        cen = rng.standard_normal((50, dim))
        pts = (cen[rng.integers(0, 50, n)] + 0.01 * rng.standard_normal((n, dim))).astype(np.float32)
        qs = (cen[rng.integers(0, 50, nq)] + 0.01 * rng.standard_normal((nq, dim))).astype(np.float32)
    else:
        pts = (rng.standard_normal((n, dim)) * 10.0 ** rng.integers(-6, 7, (n, 1))).astype(np.float32)
        qs = (rng.standard_normal((nq, dim)) * 10.0 ** rng.integers(-6, 7, (nq, 1))).astype(np.float32)
    tree = pn.BallTree.new(pts, pn.distance.Cosine())
    assert tree.bf16_eligible
    gi, gd = tree.query_batch(qs, k)
    st = tree.stats()
    tree.set_engine("exact")
    ei, ed = tree.query_batch(qs, k)
    assert _same_dist(gd, ed) and np.array_equal(gi, ei), family
    oi, od = _cosine_oracle_large(oracle_mod, pts, qs[:6], k, shortlist=2000 if family in ("clustered", "positive") else 400)
    assert _same_dist(gd[:6], od) and np.array_equal(gi[:6], oi), family
    print(f"{family}: unproven {st['fallback_queries']} of {nq}, candidates per query {st['candidates'] / st['queries']:.0f}")
    if family in ("gaussian", "scaled"):
        assert st["fallback_queries"] <= nq // 20, st
