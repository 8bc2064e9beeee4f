"""f64 indexes behind the bf16 MFMA filter (the reference is generic over A, its own harness is f64).

The first tier's bound is a statement about real vectors -- L'(q, p) <= |q-p|^2 - |q|^2 with every constant computed
in f64 from the index's own f64 coordinates -- so an f64 index gets the same bf16 tile images as an f32 one; what
follows the filter is f64: the candidates' distances in the reference's sequential unfused fold (src/distance.rs:26-35),
64-bit keys, the proof with u = 2^-53, and the exact f64 engine for the queries it cannot prove.  Every answer must be
bit-identical to the oracle's f64 brute force, whatever tier produced it."""
import numpy as np
import pytest

from conftest import uniform

pytestmark = pytest.mark.gpu


def _f64(shape, seed):
    return np.random.default_rng(seed).random(shape)  # 53 random bits: not representable in f32


@pytest.mark.parametrize("n,dim,nq,k", [(20000, 128, 300, 10), (5000, 33, 100, 5), (70000, 96, 520, 40),
                                        (9000, 17, 64, 1), (8192, 300, 100, 10), (4100, 8, 50, 3)])
def test_f64_index_through_the_bf16_tier(pn, oracle_mod, n, dim, nq, k):
    pts, qs = _f64((n, dim), 41 + n), _f64((nq, dim), 42 + n)
    qs[:3] = pts[:3]  # self-queries: distance exactly 0
    want_i, want_d = oracle_mod.brute_knn(pts, qs, k)
    tree = pn.BallTree.euclidean(pts)
    assert tree.bf16_eligible
    for eng in ("bf16", "auto", "exact"):
        tree.set_engine(eng)
        tree.stats(reset=True)
        idx, dist = tree.query_batch(qs, k)
        assert dist.dtype == np.float64
        assert np.array_equal(idx, want_i), eng
        assert dist.tobytes() == want_d.tobytes(), eng
        if eng == "bf16":
            st = tree.stats()
            assert st["candidates"] > 0 and st["fallback_queries"] <= nq // 20  # the filter tier really served it


def test_f64_unproven_queries_take_the_f64_exact_engine(pn, oracle_mod):
    """Ties beyond k' (many equal rows), a NaN query, an f64 query far outside the f32 range: the first tier cannot
    prove them, the f64 exact engine answers them in place, bit-identical to the oracle."""
    n, dim, nq, k = 30000, 64, 200, 10
    pts, qs = _f64((n, dim), 51), _f64((nq, dim), 52)
    pts[1000:1400] = pts[1000]          # 400 equal rows: equal distances far beyond any k'
    qs[0] = pts[1000] + 1e-9            # its neighbours are those rows
    qs[1, 3] = np.nan
    qs[2] = 1e200
    tree = pn.BallTree.euclidean(pts)
    tree.set_engine("bf16")
    idx, dist = tree.query_batch(qs, k)
    want_i, want_d = oracle_mod.brute_knn(pts, qs, k)
    assert np.array_equal(idx, want_i)
    assert dist.tobytes() == want_d.tobytes()
    assert tree.stats()["fallback_queries"] >= 3


def test_f64_values_that_differ_only_below_f32_precision(pn, oracle_mod):
    """Rows that are equal as f32 and differ in the 40th bit: the bf16 bound cannot separate them (it need not), the f64
    re-rank must -- the order of the answers follows the f64 distances."""
    n, dim, k = 8000, 32, 6
    base = _f64((n, dim), 61)
    pts = base.copy()
    pts[10:16] = pts[10] + np.arange(6)[:, None] * 2.0 ** -40   # six near-copies of one row
    q = pts[10:11] - 2.0 ** -42
    tree = pn.BallTree.euclidean(pts)
    tree.set_engine("bf16")
    idx, dist = tree.query_batch(q, k)
    want_i, want_d = oracle_mod.brute_knn(pts, q, k)
    assert np.array_equal(idx, want_i) and dist.tobytes() == want_d.tobytes()
    assert idx[0].tolist() == [10, 11, 12, 13, 14, 15]


def test_f64_device_entry_point(pn, oracle_mod):
    import torch
    n, dim, nq, k = 50000, 128, 1000, 10
    pts, qs = _f64((n, dim), 71), _f64((nq, dim), 72)
    tree = pn.BallTree.euclidean(pts)
    qd = torch.from_numpy(qs).to("cuda:0")
    di, dd = tree.query_device(qd, k)
    torch.cuda.synchronize()
    want_i, want_d = oracle_mod.brute_knn(pts, qs[:100], k)
    assert np.array_equal(di.cpu().numpy()[:100].astype(np.uint64), want_i)
    assert dd.cpu().numpy()[:100].tobytes() == want_d.tobytes()
    tree.set_engine("exact")
    ei, ed = tree.query_device(qd, k)
    torch.cuda.synchronize()
    assert torch.equal(di, ei) and torch.equal(dd.view(torch.int64), ed.view(torch.int64))


@pytest.mark.parametrize("n,dim,nq", [(30000, 64, 150), (12000, 128, 90), (9000, 300, 40)])
def test_f64_query_radius_through_the_bf16_tier(pn, oracle_mod, n, dim, nq):
    """BallTree::query_radius on an f64 index: the bf16 filter against each query's fixed bound (threshold rounded with
    u = 2^-53), the survivors checked in the reference's f64 fold with the strict '<' of src/ball_tree.rs:277."""
    pts, qs = _f64((n, dim), 81 + n), _f64((nq, dim), 82 + n)
    qs[:2] = pts[:2]
    tree = pn.BallTree.euclidean(pts)
    _, d = oracle_mod.brute_knn(pts, qs, 4)
    radii = [float(np.median(d[:, -1])), float(d[:, 1].max()) * 1.0000001, float(d[5, 2])]  # incl. a radius EQUAL to a distance
    for eng in ("bf16", "auto", "exact"):
        tree.set_engine(eng)
        for r in radii:
            off, ids = tree.query_radius_batch(qs, r)
            assert off[0] == 0 and off[-1] == len(ids)
            for a in range(nq):
                want = oracle_mod.brute_radius(pts, qs[a], np.float64(r))
                assert np.array_equal(ids[int(off[a]):int(off[a + 1])], want), (eng, r, a)


@pytest.mark.parametrize("k", [10, 100])
def test_f64_second_tier_many_segment_path(pn, oracle_mod, k):
    """A corpus of >= 65 536 rows: the first 256 unproven queries of a call are scanned over up to 512 row segments and
    selected in two levels (f64 keys, the merge kernel on (f64 distance, index) pairs), the rest in the rounds of the
    exact engine.  With k' = k a good part of the batch is unproven; with duplicated rows only a handful."""
    from petal_neighbors_amd import _lib
    pts, qs = _f64((90000, 32), 61), _f64((700, 32), 62)
    tree = pn.BallTree.euclidean(pts)
    tree.set_engine("bf16")
    tree.set_option(_lib.PN_OPT_FILTER_SLOTS, k)
    tree.set_option(_lib.PN_OPT_SEGMENTS, 1)
    idx, dist = tree.query_batch(qs, k)
    want_i, want_d = oracle_mod.brute_knn(pts, qs, k)
    assert np.array_equal(idx, want_i)
    assert dist.tobytes() == want_d.tobytes()
    assert tree.stats()["fallback_queries"] > 256   # both the many-segment path and the rounds behind it ran
    pts2 = pts.copy()
    pts2[1000:1040] = pts2[7]
    q2 = np.concatenate([pts2[7:8] + 1e-12, qs[:100]])
    tree2 = pn.BallTree.euclidean(pts2)
    idx, dist = tree2.query_batch(q2, 10)
    want_i, want_d = oracle_mod.brute_knn(pts2, q2, 10)
    assert np.array_equal(idx, want_i)
    assert dist.tobytes() == want_d.tobytes()
    assert 1 <= tree2.stats()["fallback_queries"] <= 8


def test_f64_index_from_rows_already_in_hbm(pn, oracle_mod):
    """pn_index_create_device_f64: an f64 index built from a device array (a strided view of a wider one) answers like the
    one built from the host copy."""
    import torch
    n, dim, nq, k = 20000, 24, 150, 7
    wide = torch.from_numpy(_f64((n, dim + 8), 71)).to("cuda:0")
    view = wide[:, :dim]                       # row stride dim + 8, inner stride 1
    tree = pn.BallTree.from_device(view)
    assert tree.dtype == np.float64
    pts = view.cpu().numpy().copy()
    qs = _f64((nq, dim), 72)
    idx, dist = tree.query_batch(qs, k)
    want_i, want_d = oracle_mod.brute_knn(pts, qs, k)
    assert np.array_equal(idx, want_i)
    assert dist.tobytes() == want_d.tobytes()
