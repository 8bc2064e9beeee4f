"""bf16 first tier (petal-neighbors_amd/csrc/bf16_filter.hip) on an MI355X.

Two kinds of test:
  * the INEQUALITY the tier's proof rests on -- L'(q,p) + qn <= |q-p|^2 with qn <= |q|^2 -- checked pair by
    pair against f64, on benign and on hostile data, together with the measured accumulation error of the
    matrix core against the allowance g = 2^-13 the proof makes;
  * END-TO-END parity: results through the C ABI with the bf16 tier forced are bit-identical to the oracle,
    whatever the tier could or could not prove (unproven queries go to the f32 tiers).
"""
import ctypes as C

import numpy as np
import pytest

from conftest import uniform

pytestmark = pytest.mark.gpu


def _bounds(pn, tree, qs, n_rows):
    from petal_neighbors_amd import _lib
    from petal_neighbors_amd.errors import check
    qs = np.ascontiguousarray(qs, dtype=np.float32)
    nq, d = qs.shape
    n_rows = min(n_rows, tree.num_points())
    out = np.empty((nq, n_rows), dtype=np.float32)
    qn = np.empty(nq, dtype=np.float64)
    mu = np.empty(d, dtype=np.float32)
    check(_lib.lib().pn_bf16_bounds_f32(tree._h, qs.ctypes.data, nq, d, d, n_rows, out.ctypes.data, qn.ctypes.data,
                                        mu.ctypes.data))
    return out, qn, mu


def _bf16_round(x):
    """round-to-nearest-even bf16 of an f32 array (as f32), with the filter's 2^-60 flush"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    y = r.astype(np.uint32).view(np.float32).reshape(np.shape(x))
    return np.where(np.abs(x) < 2.0 ** -60, np.float32(0), y)


CASES = {
    "uniform": lambda n, d, s: uniform((n, d), s),
    "centered": lambda n, d, s: uniform((n, d), s) - np.float32(0.5),
    "offset1000": lambda n, d, s: uniform((n, d), s) + np.float32(1000.0),
    "scaled1e6": lambda n, d, s: (uniform((n, d), s) - np.float32(0.5)) * np.float32(1e6),
    "tiny1e-20": lambda n, d, s: (uniform((n, d), s) - np.float32(0.5)) * np.float32(1e-20),
    "sparse": lambda n, d, s: np.where(uniform((n, d), s + 7) < 0.9, np.float32(0), uniform((n, d), s)).astype(np.float32),
    "mixed_scales": lambda n, d, s: ((uniform((n, d), s) - np.float32(0.5))
                                      * (np.float32(10.0) ** (np.arange(d, dtype=np.float32) % 9 - 4))).astype(np.float32),
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("dim", [8, 33, 96, 110, 128])
def test_lower_bound_inequality(pn, name, dim):
    n, nq = 4096, 96
    pts = CASES[name](n, dim, 11)
    qs = CASES[name](nq, dim, 12)
    tree = pn.BallTree.euclidean(pts)
    assert tree.bf16_eligible
    L, qn, mu = _bounds(pn, tree, qs, 1024)
    p64, q64 = pts[:1024].astype(np.float64), qs.astype(np.float64)
    qq = ((q64 - mu.astype(np.float64)) ** 2).sum(1)  # the tier works with vectors translated by the corpus mean
    assert np.all(qn <= qq)
    if tree.bf16_layout == 1:
        assert np.all(qn >= qq * (1 - 1e-11))
    else:  # layout 2: the per-query error constant E(q) is already subtracted (bf16_filter.hip, bf16_ci_dim)
        assert tree.bf16_layout == 2 and dim % 16 in (0, 12, 13, 14, 15) and dim >= 17
        assert np.all(qn >= qq - 0.05 * qq - 1e-30)
    # the translation is the corpus mean where that shrinks the squared norms at least 16x, else none
    assert np.all(mu == 0) or np.allclose(mu, pts.astype(np.float64).mean(0), rtol=1e-5, atol=1e-30)
    if name == "offset1000":
        assert np.all(mu > 999)
    d2 = ((q64[:, None, :] - p64[None, :, :]) ** 2).sum(2)
    assert np.all(np.isfinite(L))
    gap = d2 - (L.astype(np.float64) + qn[:, None])
    assert gap.min() >= 0.0, f"{name}/D={dim}: bound exceeds the squared distance by {-gap.min()}"
    # the bound is useful on benign data: slack small against the spread of the squared distances
    if name in ("uniform", "centered", "offset1000"):
        assert gap.max() < 0.05 * d2.mean() + 1e-3


@pytest.mark.parametrize("name", ["centered", "offset1000", "mixed_scales", "uniform"])
@pytest.mark.parametrize("dim", [16, 100, 128])
def test_matrix_core_accumulation_error_is_far_inside_the_allowance(pn, name, dim):
    """The one hardware assumption of the bound: |value delivered by v_mfma_f32_32x32x16_bf16 - exact sum of its
    bf16 x bf16 terms| <= g * sum|terms| with g = 2^-13.  The terms are rebuilt bit for bit on the host
    (tests/test_bf16_bound_model.py holds the same constants and roundings as the pack kernels), summed in f64, and
    compared with what the GPU returned."""
    from test_bf16_bound_model import corpus_columns, query_columns, G
    n, nq = 1024, 64
    pts = CASES[name](n, dim, 21)
    qs = CASES[name](nq, dim, 22)
    tree = pn.BallTree.euclidean(pts)
    L, _, mu = _bounds(pn, tree, qs, n)
    ph, pieces, bp, dp = corpus_columns(pts, mu)
    mq, aq, cq, _ = query_columns(qs, mu)
    if tree.bf16_layout == 2:
        # the chain is the row norm (one f32, the accumulator's initial value) + the data products, nothing else
        from test_bf16_bound_model import corpus_norm_f32
        cn = corpus_norm_f32(pts, mu)
        exact = cn[None, :] + mq @ ph.T
        mags = cn[None, :] + np.abs(mq) @ np.abs(ph).T
    else:
        exact = pieces.sum(1)[None, :] + mq @ ph.T - aq[:, None] * bp[None, :] - cq[:, None] * dp[None, :]
        mags = pieces.sum(1)[None, :] + np.abs(mq) @ np.abs(ph).T + aq[:, None] * bp[None, :] + cq[:, None] * dp[None, :]
    ratio = np.abs(L.astype(np.float64) - exact) / (G * mags)
    print(f"{name}/D={dim}: accumulation error / allowance: max {ratio.max():.4f}, mean {ratio.mean():.5f}")
    assert ratio.max() < 0.02, f"matrix-core accumulation error uses {ratio.max():.4f} of the allowance"  # measured: <= 0.0016


def _check(pn, oracle_mod, pts, qs, k, opts=None, expect_fallback=None):
    from petal_neighbors_amd import _lib
    tree = pn.BallTree.euclidean(pts)
    assert tree.bf16_eligible
    tree.set_engine("bf16")
    for o, v in (opts or {}).items():
        tree.set_option(o, v)
    idx, dist = tree.query_batch(qs, k)
    oidx, odist = oracle_mod.brute_knn(pts, qs, k)
    assert dist.tobytes() == odist.tobytes(), "distances differ"
    assert np.array_equal(idx, oidx), "indices differ"
    st = tree.stats()
    if expect_fallback is not None:
        assert (st["fallback_queries"] > 0) == expect_fallback, st
    return tree, st


@pytest.mark.parametrize("n,dim,nq,k", [(5000, 128, 300, 10), (20000, 96, 700, 1), (9000, 64, 257, 25),
                                          (4100, 8, 130, 3), (30000, 128, 64, 100), (6000, 33, 500, 7),
                                          (4096, 17, 1, 5), (70000, 128, 1000, 10)])
def test_bf16_engine_parity(pn, oracle_mod, n, dim, nq, k):
    pts = uniform((n, dim), 100 + dim)
    qs = uniform((nq, dim), 200 + dim)
    _check(pn, oracle_mod, pts, qs, k)


def test_bf16_engine_is_the_auto_default_and_proves_uniform_data(pn, oracle_mod):
    pts = uniform((50000, 128), 5)
    qs = uniform((2000, 128), 6)
    tree = pn.BallTree.euclidean(pts)
    idx, dist = tree.query_batch(qs, 10)  # engine auto
    oidx, odist = oracle_mod.brute_knn(pts, qs, 10)
    assert dist.tobytes() == odist.tobytes() and np.array_equal(idx, oidx)
    st = tree.stats()
    assert st["fallback_queries"] <= 20, st
    assert 10 <= st["candidates"] / st["queries"] <= 2000, st


def test_bf16_common_offset_is_served_by_the_tier(pn, oracle_mod):
    # the tier works with vectors translated by the corpus mean: a large common offset costs nothing
    pts = uniform((8000, 64), 31) + np.float32(1000.0)
    qs = uniform((200, 64), 32) + np.float32(1000.0)
    _check(pn, oracle_mod, pts, qs, 5, expect_fallback=False)


def test_bf16_hostile_data_falls_back_and_stays_exact(pn, oracle_mod):
    # tight clusters far apart: distances inside a cluster are far below bf16's resolution of the coordinates,
    # bf16 resolves nothing and every query must be handed to the f32 tiers
    rng = np.random.default_rng(9)
    cen = (rng.random((8, 64), dtype=np.float32) * 100).astype(np.float32)
    pts = (cen[rng.integers(0, 8, 8000)] + 0.01 * rng.standard_normal((8000, 64), dtype=np.float32)).astype(np.float32)
    qs = (pts[rng.integers(0, 8000, 200)] + 0.001 * rng.standard_normal((200, 64), dtype=np.float32)).astype(np.float32)
    _check(pn, oracle_mod, pts, qs, 5, expect_fallback=True)


def test_bf16_ties_and_duplicates(pn, oracle_mod):
    base = uniform((3000, 32), 41)
    pts = np.concatenate([base, base[:1500], base[:700]]).astype(np.float32)  # exact duplicates -> distance ties
    qs = np.concatenate([base[:64], uniform((64, 32), 42)]).astype(np.float32)
    _check(pn, oracle_mod, pts, qs, 12)


def test_bf16_clustered_order(pn, oracle_mod):
    # corpus sorted by cluster: all neighbours of a query sit in one segment
    rng = np.random.default_rng(3)
    centers = rng.random((16, 48), dtype=np.float32) * 10
    pts = np.concatenate([c + 0.05 * rng.standard_normal((1000, 48), dtype=np.float32) for c in centers]).astype(np.float32)
    qs = (centers[rng.integers(0, 16, 300)] + 0.05 * rng.standard_normal((300, 48), dtype=np.float32)).astype(np.float32)
    _check(pn, oracle_mod, pts, qs, 10)


def test_bf16_small_slots_force_fallback(pn, oracle_mod):
    from petal_neighbors_amd import _lib
    pts = uniform((20000, 128), 51)
    qs = uniform((256, 128), 52)
    # k' = k and a single segment: the k-th bound can never clear the k-th distance
    _check(pn, oracle_mod, pts, qs, 10, opts={_lib.PN_OPT_FILTER_SLOTS: 10, _lib.PN_OPT_SEGMENTS: 1},
           expect_fallback=True)


def test_bf16_nonfinite_query_and_large_k(pn, oracle_mod):
    pts = uniform((6000, 40), 61)
    qs = uniform((70, 40), 62)
    qs[3, 5] = np.nan
    qs[9, 0] = np.inf
    qs[11, :] = 1e25
    tree = pn.BallTree.euclidean(pts)
    tree.set_engine("bf16")
    for k in (4, 150):
        idx, dist = tree.query_batch(qs, k)
        oidx, odist = oracle_mod.brute_knn(pts, qs, k)
        na, nb = np.isnan(dist), np.isnan(odist)
        assert np.array_equal(na, nb) and dist[~na].tobytes() == odist[~nb].tobytes()
        ok = ~np.isnan(odist).any(axis=1) & ~np.isinf(odist).any(axis=1)
        assert np.array_equal(idx[ok], oidx[ok])


def test_bf16_ineligible_index(pn):
    pts = uniform((5000, 16), 71)
    pts[17, 3] = np.inf
    tree = pn.BallTree.euclidean(pts)
    assert not tree.bf16_eligible
    with pytest.raises(Exception):
        tree.set_engine("bf16")
    wide = pn.BallTree.euclidean(uniform((5000, 1030), 72))  # (round 4: the tier serves rows up to 4096 columns)
    assert wide.bf16_eligible
    wider = pn.BallTree.euclidean(uniform((600, 4100), 73))  # rows beyond 4096 columns: no bf16 tier
    assert not wider.bf16_eligible


def test_bf16_radius_mixed_density_falls_back_per_query(pn, oracle_mod):
    """A few queries sit in a dense clump (thousands of rows within the radius: their survivor lists overflow),
    the others have sparse neighbourhoods: only the dense ones are re-run exactly, the call stays on the tier."""
    rng = np.random.default_rng(17)
    base = uniform((40000, 32), 81)
    clump = (base[123] + 0.01 * rng.standard_normal((3000, 32))).astype(np.float32)
    pts = np.concatenate([base, clump]).astype(np.float32)
    qs = np.concatenate([uniform((60, 32), 82), clump[:3] + np.float32(0.001)]).astype(np.float32)
    tree = pn.BallTree.euclidean(pts)
    _, d = oracle_mod.brute_knn(pts, qs[:60], 4)
    r = float(np.median(d[:, 3]))  # sparse queries: a handful of rows; clump queries: ~3000
    off, idx = tree.query_radius_batch(qs, r)
    sizes = []
    for a in range(len(qs)):
        want = oracle_mod.brute_radius(pts, qs[a], np.float32(r))
        assert np.array_equal(idx[int(off[a]):int(off[a + 1])], want), a
        sizes.append(len(want))
    assert max(sizes) > 2500 and sorted(sizes)[len(sizes) // 2] < 50
    st = tree.stats()
    assert 1 <= st["fallback_queries"] <= 5, st  # the clump queries only, not the whole call


def test_bf16_radius_boundary_on_tight_bounds(pn, oracle_mod):
    """Radius thresholds vs TAGGED bounds: coordinates that bf16 represents exactly (multiples of 1/8 in
    [0, 4)) make the filter's bound as tight as it gets -- e_p = e_q = 0, only the g-terms remain -- so rows sit as
    close to the per-query threshold as the arithmetic allows.  With r equal to the exact distance of each query's
    j-th neighbour (that row excluded by the strict '<', all nearer ones included) and one ulp above it (included),
    engine bf16 must return exactly the oracle's sets."""
    rng = np.random.default_rng(5)
    for n, dim in ((20000, 16), (12000, 128)):
        pts = (rng.integers(0, 32, size=(n, dim)) / 8.0).astype(np.float32)
        qs = (rng.integers(0, 32, size=(96, dim)) / 8.0).astype(np.float32)
        tree = pn.BallTree.euclidean(pts)
        tree.set_engine("bf16")
        _, d = oracle_mod.brute_knn(pts, qs, 12)
        for j in (0, 3, 11):
            for bump in (0, 1):
                # one radius per call: take query a's own j-th distance for every a by calling per group of equal r
                rs = d[:, j] if not bump else np.nextafter(d[:, j], np.float32(np.inf))
                for r in np.unique(rs)[:24]:
                    off, idx = tree.query_radius_batch(qs, np.float32(r))
                    for a in np.nonzero(rs == r)[0][:4]:
                        want = oracle_mod.brute_radius(pts, qs[a], np.float32(r))
                        assert np.array_equal(idx[int(off[a]):int(off[a + 1])], want), (n, dim, j, bump, float(r), int(a))


def test_heterogeneous_norms_keep_the_per_row_layout(pn, oracle_mod):
    """The norm-in-the-accumulator layout replaces each row's error constants by the corpus maxima: rows of very
    different norms would make that bound useless, so such a corpus keeps the five per-row columns (layout 1) --
    and answers stay exact either way."""
    rng = np.random.default_rng(9)
    pts = uniform((20000, 128), 31) * (10.0 ** rng.integers(-2, 3, size=(20000, 1))).astype(np.float32)
    qs = pts[rng.integers(0, 20000, size=200)] + uniform((200, 128), 32) * np.float32(0.01)
    tree = pn.BallTree.euclidean(pts)
    assert tree.bf16_eligible and tree.bf16_layout == 1
    assert pn.BallTree.euclidean(uniform((20000, 128), 33)).bf16_layout == 2
    tree.set_engine("bf16")
    idx, dist = tree.query_batch(qs, 10)
    oi, od = oracle_mod.brute_knn(pts, qs, 10)
    assert dist.tobytes() == od.tobytes() and np.array_equal(idx, oi)


@pytest.mark.parametrize("k", [10, 100])
def test_second_tier_many_segment_path(pn, oracle_mod, k):
    """The device-driven second tier: with k' = k the first tier's proofs fail for a good part of the queries; the
    first 256 flagged ones of the call take the many-segment scan with the two-level selection (corpus >= 65 536
    rows), the rest the rounds of the ordinary exact engine -- answers are the oracle's either way."""
    from petal_neighbors_amd import _lib
    pts = uniform((90000, 64), 41)
    qs = uniform((1500, 64), 42)
    tree, st = _check(pn, oracle_mod, pts, qs, k, opts={_lib.PN_OPT_FILTER_SLOTS: k, _lib.PN_OPT_SEGMENTS: 1})
    assert st["fallback_queries"] > 300, st
    # and a handful only: ties beyond k' on duplicated rows
    pts2 = pts.copy()
    pts2[1000:1040] = pts2[7]
    q2 = np.concatenate([pts2[7:8], qs[:200]])
    tree, st = _check(pn, oracle_mod, pts2, q2, 10)
    assert 1 <= st["fallback_queries"] <= 8, st


def test_narrow_partition_edge_cases(pn, oracle_mod):
    """the software-pipelined main loop (three LDS buffers, barrier in mid-tile) on runs of every length: few row
    tiles, the segment cap, and persistent slices that straddle query tiles (more query tiles than workgroups), whose
    runs begin and end anywhere -- including runs of one and two tiles"""
    from petal_neighbors_amd import _lib
    # few row tiles: 66 tiles over up to 2 workgroups per query tile
    _check(pn, oracle_mod, uniform((4200, 48), 31), uniform((300, 48), 32), 5)
    # the segment cap with several query tiles (aligned: 3 workgroups per query tile)
    _check(pn, oracle_mod, uniform((30000, 128), 33), uniform((1000, 128), 34), 10, opts={_lib.PN_OPT_SEGMENTS: 3})
    # more query tiles than workgroup slots, persistent equal slices (the option switches the grid-in-rounds plan off):
    # slices straddle query tiles, runs are cut wherever a slice ends
    n, dim, nq, k = 4500, 16, 140000, 5
    pts, qs = uniform((n, dim), 35), uniform((nq, dim), 36)
    tree = pn.BallTree.euclidean(pts)
    tree.set_engine("bf16")
    tree.set_option(_lib.PN_OPT_SEGMENTS, 2)
    idx, dist = tree.query_batch(qs, k)
    ref = pn.BallTree.euclidean(pts)
    ref.set_engine("exact")
    ridx, rdist = ref.query_batch(qs, k)
    assert dist.tobytes() == rdist.tobytes() and np.array_equal(idx, ridx)
    sel = np.random.default_rng(37).choice(nq, 300, replace=False)
    oidx, odist = oracle_mod.brute_knn(pts, qs[sel], k)
    assert dist[sel].tobytes() == odist.tobytes() and np.array_equal(idx[sel], oidx)


def test_shared_thresholds_only_change_the_candidate_count(pn, oracle_mod):
    """Shared thresholds (bf16_filter.hip): refresher workgroups lower every segment's threshold to the r-th smallest bound
    of the union of what the segments of a query hold, while the filter runs.  Any such threshold is valid -- the
    answers must be bit-identical with the option off, on, and with a rank so small (r = 2) that the thresholds drop
    below the true neighbours and nearly every query has to be answered by the next tier."""
    from petal_neighbors_amd import _lib
    # 12 query tiles x 32 segments = 384 main workgroups of 586 row tiles each (shared thresholds need runs of >= 512
    # tiles: a refresher pass takes ~0.25 ms), refreshers behind them
    n, dim, nq, k = 1_200_000, 128, 3000, 10
    pts, qs = uniform((n, dim), 3101), uniform((nq, dim), 3102)
    want_i, want_d = oracle_mod.brute_knn(pts, qs[:200], k)
    tree = pn.BallTree.euclidean(pts)
    tree.set_engine("bf16")
    got = {}
    for mode in (0, 1, 44, 2):
        if mode == 2:  # a fresh index: a call that leaves > 1/16 of its queries unproven makes an index plan conservatively
            tree = pn.BallTree.euclidean(pts)  # (sticky), and r = 2 does exactly that -- so does r = 24 for a third of them
            tree.set_engine("bf16")
        # (thresholds from the scout launch: with the index's seed model the default plan does not share at all --
        # tests/test_gpu_seed_model.py covers the model, with and without an explicit sharing rank)
        tree.set_option(_lib.PN_OPT_SEED_MODEL, 0)
        tree.set_option(_lib.PN_OPT_SHARED_THRESHOLDS, mode)
        tree.stats(reset=True)
        for _ in range(3):  # consecutive calls on one workspace: the epoch of the published words changes every call
            idx, dist = tree.query_batch(qs, k)
        st = tree.stats()
        got[mode] = (idx, dist, st["candidates"] / st["queries"], st["fallback_queries"])
        assert np.array_equal(idx[:200], want_i) and dist[:200].tobytes() == want_d.tobytes(), mode
        assert np.array_equal(idx, got[0][0]) and dist.tobytes() == got[0][1].tobytes(), mode
    assert got[0][3] <= 3 and got[1][3] <= 3          # benign data: (nearly) everything proven, with or without
    assert got[1][2] < got[0][2]                      # the default rank keeps fewer candidates than no sharing
    assert got[2][3] > nq // 2                        # r = 2: most queries of its first call went to the next tier


@pytest.mark.parametrize("family", ["uniform", "duplicated", "clustered"])
def test_shared_thresholds_below_the_kth_bound_never_prove_a_wrong_answer(pn, oracle_mod, family):
    """a main wave adopts a shared word S (tau = min(tau, S)) and from then on drops rows with bounds
    >= S; a LATER compaction of its buffer used to overwrite tau with the buffer's k'-th key T, which exceeds S whenever
    the buffer holds fewer than k' entries below S (entries appended before S was adopted), and T was what the proof saw
    -- rows with bounds in [S, T) were gone and the query still counted as proven.  k = 40 plans k' ~ 48 in 128-slot
    buffers that are cut to k' at the end of a run whenever they hold more; ranks from far below k to about the number
    of rows whose bound lies under the k-th distance (~100) put S below the k-th neighbour's bound for anything from
    all to a good share of the queries, with buffers that do fill beyond k' at the larger ranks.  Thresholds are now
    monotone (min with T at every compaction): such queries fail their proof and the next tier answers them.  Whatever
    happens in between, answers are the oracle's bit for bit."""
    from petal_neighbors_amd import _lib
    rng = np.random.default_rng(7100)
    n, dim, nq, k = 220_000, 64, 520, 40
    if family == "uniform":
        pts, qs = uniform((n, dim), 7107), uniform((nq, dim), 7108)
    elif family == "duplicated":  # every row eight times (ties beyond k' as well), queries near rows
        base = uniform((n // 8, dim), 7102)
        pts = np.tile(base, (8, 1))
        qs = (base[rng.integers(0, n // 8, nq)] + np.float32(0.02) * uniform((nq, dim), 7103)).astype(np.float32)
    else:  # 200 loose clusters: several times the usual number of rows with bounds under the 40th neighbour's distance
        cen = uniform((200, dim), 7101)
        pts = (cen[rng.integers(0, 200, n)] + 0.12 * rng.standard_normal((n, dim))).astype(np.float32)
        qs = (cen[rng.integers(0, 200, nq)] + 0.12 * rng.standard_normal((nq, dim))).astype(np.float32)
    pts = np.ascontiguousarray(pts[rng.permutation(n)])
    want_i, want_d = oracle_mod.brute_knn(pts, qs, k)
    seen = {}
    for rank in (8, 40, 60, 80, 110, 1):
        tree = pn.BallTree.euclidean(pts)   # (a fresh index per rank: many unproven queries make an index plan conservatively)
        assert tree.bf16_eligible
        tree.set_engine("bf16")
        tree.set_option(_lib.PN_OPT_SEGMENTS, 4)           # 4 segments x 859-tile runs: shared thresholds are planned
        tree.set_option(_lib.PN_OPT_SEED_MODEL, 0)         # (scouted starting thresholds, as when the defect was found)
        tree.set_option(_lib.PN_OPT_SHARED_THRESHOLDS, rank)
        idx, dist = tree.query_batch(qs, k)
        st = tree.stats()
        seen[rank] = (st["fallback_queries"], round(st["candidates"] / st["queries"], 1))
        assert dist.tobytes() == want_d.tobytes(), (family, rank, "distances differ", seen)
        # rows with exactly equal distances (duplicates) come back in ascending index, as the oracle orders them
        assert np.array_equal(idx, want_i), (family, rank, "indices differ", seen)
    print(f"{family}: (unproven queries, candidates per query) by rank: {seen}")
    if family == "uniform":
        assert seen[8][0] > nq // 2, seen      # the shared word did cut below the k-th neighbour: sharing was active


def test_large_k_takes_a_sample_sized_by_its_relevant_rows(pn, oracle_mod):
    """k = 100: the shared scout's sample holds ~9 of the ~244 relevant rows (lambda = R / 27), the seed is a rank beyond
    the 12 a lane's list holds (merged by the seed kernel, not the main launch's prologue), and buffers of 128 slots
    are planned for twice the mean count.  Whatever the seed, the answers are the oracle's."""
    n, dim, nq, k = 200_000, 32, 600, 100
    pts, qs = uniform((n, dim), 4101), uniform((nq, dim), 4102)
    tree, st = _check(pn, oracle_mod, pts, qs, k)
    assert st["fallback_queries"] <= nq // 50, st       # the tier served the batch
    assert st["candidates"] / st["queries"] >= k, st


def test_radius_reruns_of_a_few_dense_queries_on_a_larger_corpus(pn, oracle_mod):
    """The handful of queries whose survivor lists overflow are re-run exactly over up to 1024 row segments (the
    two-pass CSR scan with per-(query, segment) offsets): same lists as the oracle's, ascending."""
    rng = np.random.default_rng(23)
    base = uniform((600_000, 16), 4201)
    clump = (base[777] + 0.002 * rng.standard_normal((4000, 16))).astype(np.float32)
    pts = np.concatenate([base, clump]).astype(np.float32)
    qs = np.concatenate([uniform((30, 16), 4202), clump[:2] + np.float32(0.0005)]).astype(np.float32)
    tree = pn.BallTree.euclidean(pts)
    _, d = oracle_mod.brute_knn(pts, qs[:30], 3)
    r = float(np.median(d[:, 2]))
    off, idx = tree.query_radius_batch(qs, r)
    for a in range(len(qs)):
        want = oracle_mod.brute_radius(pts, qs[a], np.float32(r))
        assert np.array_equal(idx[int(off[a]):int(off[a + 1])], want), a
    st = tree.stats()
    assert 1 <= st["fallback_queries"] <= 4, st


def test_library_checks_the_matrix_cores_accumulation_itself(pn):
    """the proof's allowance g = 2^-13 for the matrix core's f32 accumulation is a measured property
    of gfx950; the guard used to be the two pytest measurements above.  Now the library contracts synthetic chains (8 and
    65 MFMA steps, mixed signs and scales, non-zero accumulator) the first time an index on a device gets its bf16 tier,
    compares with the terms rebuilt in f64, and refuses the tier above 2 % of the allowance.  pn_bf16_selftest reruns
    that check and reports the fraction."""
    from petal_neighbors_amd import _lib
    from petal_neighbors_amd.errors import check
    r = C.c_float(-1.0)
    check(_lib.lib().pn_bf16_selftest(0, C.byref(r)))
    print(f"matrix-core accumulation self-test: {r.value:.5f} of the allowance")
    assert 0.0 <= r.value < 0.02, r.value            # measured by the pytest twins: <= 0.0016
    assert r.value > 0.0                              # it did contract something (f32 accumulation is not exact)
    tree = pn.BallTree.euclidean(uniform((4096, 64), 77))
    assert tree.bf16_eligible                         # and the verdict let the tier through


@pytest.mark.parametrize("k", [300, 600, 1000])
def test_large_k_is_served_by_the_tier_with_more_segments(pn, oracle_mod, k):
    """Round 4: k beyond what 12 segments' buffers hold (k' = R / segments + margin <= 224) is planned with MORE, shorter
    segments per query tile (a grid run in rounds).  Round 3 split the rows into parts instead -- which does not add
    workgroups -- planned k' for twice the segments a query really had, and 97 % of a k = 500 batch failed its proof (the
    index then switched the tier off: 126 ms per 10^4 queries on the exact engine against 5 ms here).  Answers are the
    oracle's whatever the plan; what is asserted on top is that the tier DID serve the batch."""
    n, dim, nq = 400_000, 64, 300
    pts, qs = uniform((n, dim), 9301), uniform((nq, dim), 9302)
    tree = pn.BallTree.euclidean(pts)
    want_i, want_d = oracle_mod.brute_knn(pts, qs, k)
    for call in range(2):   # (a second call: the plan did not turn conservative after the first)
        tree.stats(reset=True)
        idx, dist = tree.query_batch(qs, k)
        st = tree.stats()
        assert dist.tobytes() == want_d.tobytes() and np.array_equal(idx, want_i), (k, call)
        assert st["candidates"] / st["queries"] >= k, (k, call, st)      # the filter tier produced the candidates
        assert st["fallback_queries"] <= nq // 20, (k, call, st)
