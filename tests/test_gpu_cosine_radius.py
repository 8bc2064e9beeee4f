"""query_radius on a Cosine index behind the bf16 MFMA filter (round 4; SURVEY.md 8 f3, src/ball_tree.rs:137-142 with
src/distance.rs:76-122).  The filter runs over the rows and queries normalised in f64 against each query's fixed bound
-- a row whose reference distance is < r has |q~ - p~|^2 < 2 (r + E') + eps1 (select.hip, cos_proof_lb read backwards) --
and only FILTERS: every row returned passed `Cosine::distance(q, p) < r` evaluated in the reference's arithmetic, and
every row dropped is proven to fail it.  So the lists must equal the exact scan's (the tier off) entry for entry, and the
oracle's scalar metric on the rows near the boundary.
"""
import numpy as np
import pytest

from conftest import uniform

pytestmark = pytest.mark.gpu


def _device_lists(tree, qs, r, capacity):
    import torch
    qd = torch.from_numpy(qs).to("cuda:0")
    offs, idx, tot = tree.query_radius_device(qd, r, capacity)
    torch.cuda.synchronize()
    return offs.cpu().numpy().astype(np.uint64), idx.cpu().numpy().astype(np.uint64), int(tot.item())


def _oracle_list(oracle_mod, pts, q, r, band=2e-4):
    """rows with Cosine::distance(q, row) < r: decided in f64 away from the boundary (the reference's float result is
    within 1e-5 of the real value for these shapes), by the oracle's scalar metric inside the band around r"""
    p64, q64 = pts.astype(np.float64), q.astype(np.float64)
    d = 1.0 - (p64 @ q64) / (np.linalg.norm(p64, axis=1) * np.linalg.norm(q64))
    sure = d < float(r) - band
    near = np.flatnonzero(np.abs(d - float(r)) <= band)
    for i in near:
        sure[i] = oracle_mod.cosine(q, pts[i]) < r
    return np.flatnonzero(sure).astype(np.uint64)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,dim,nq", [(200_000, 128, 600), (60_000, 24, 300), (50_000, 200, 260)])
def test_cosine_radius_tier_equals_the_exact_scan_and_the_oracle(pn, oracle_mod, dtype, n, dim, nq):
    pts = uniform((n, dim), 8100 + dim, dtype) - dtype(0.3)
    qs = np.concatenate([pts[40:44] * dtype(1.5), uniform((nq - 4, dim), 8200 + dim, dtype) - dtype(0.3)])  # 4 parallel to rows
    tree = pn.BallTree.new(pts, pn.distance.Cosine())
    assert tree.bf16_eligible
    _, d3 = tree.query_batch(qs[4:260], 3)
    r = dtype(np.median(d3[:, 2]))                      # about half of the queries hold two or three rows
    assert 0 < r < 1
    tree.set_engine("exact")
    want_off, want_idx = tree.query_radius_batch(qs, r)
    total = int(want_off[-1])
    assert total > nq // 2
    for eng in ("bf16", "auto"):
        tree.set_engine(eng)
        tree.stats(reset=True)
        off, idx = tree.query_radius_batch(qs, r)
        assert np.array_equal(off, want_off) and np.array_equal(idx, want_idx), (eng, "host entry")
        doff, didx, dtot = _device_lists(tree, qs, r, total + 5)
        assert dtot == total and np.array_equal(doff, want_off) and np.array_equal(didx[:total], want_idx), (eng, "device entry")
    for a in (0, 1, 5, nq - 1):                         # the oracle's scalar Cosine::distance around the boundary
        assert np.array_equal(want_idx[int(want_off[a]):int(want_off[a + 1])], _oracle_list(oracle_mod, pts, qs[a], r)), a
    # a tiny radius: only rows parallel to the query (distances a few ulp around zero, some below it)
    r0 = dtype(1e-6)
    tree.set_engine("exact")
    w_off, w_idx = tree.query_radius_batch(qs, r0)
    assert int(w_off[4]) >= 4                          # the four parallel queries found their rows
    tree.set_engine("bf16")
    off, idx = tree.query_radius_batch(qs, r0)
    assert np.array_equal(off, w_off) and np.array_equal(idx, w_idx)
    tree.close()


def test_cosine_radius_hostile_inputs_keep_the_exact_answers(pn, oracle_mod):
    """a dense clump (survivor lists overflow: those queries are re-run exactly), a zero query and a NaN query (empty
    lists: the reference's distance is NaN), queries shorter and longer than the rows and r >= 1 (the exact scan serves
    them): always the exact engine's lists"""
    rng = np.random.default_rng(83)
    n, dim = 150_000, 32
    base = uniform((n, dim), 8301, np.float32) - np.float32(0.5)
    clump = (base[999] * np.float32(2.0) + np.float32(1e-3) * rng.standard_normal((3000, dim))).astype(np.float32)
    pts = np.concatenate([base, clump]).astype(np.float32)
    qs = np.concatenate([uniform((250, dim), 8302, np.float32) - np.float32(0.5), clump[:3]]).astype(np.float32)
    qs[7] = 0
    qs[9, 3] = np.nan
    tree = pn.BallTree.new(pts, pn.distance.Cosine())
    assert tree.bf16_eligible
    _, d3 = tree.query_batch(qs[20:120], 3)
    r = np.float32(np.median(d3[:, 2]))
    tree.set_engine("exact")
    want_off, want_idx = tree.query_radius_batch(qs, r)
    total = int(want_off[-1])
    assert total > 3000 and want_off[8] == want_off[7] and want_off[10] == want_off[9]
    tree.set_engine("bf16")
    off, idx = tree.query_radius_batch(qs, r)
    assert np.array_equal(off, want_off) and np.array_equal(idx, want_idx)
    doff, didx, dtot = _device_lists(tree, qs, r, total)
    assert dtot == total and np.array_equal(doff, want_off) and np.array_equal(didx[:total], want_idx)
    for a in (0, 251):
        assert np.array_equal(want_idx[int(want_off[a]):int(want_off[a + 1])], _oracle_list(oracle_mod, pts, qs[a], r)), a
    # other query lengths and a radius of half the sphere: the exact scan, whatever the engine setting
    good = np.ascontiguousarray(qs[20:60])
    for variant, rr in ((good[:, : dim - 5], r), (np.concatenate([good, good[:, :3]], axis=1), r), (good, np.float32(1.2))):
        variant = np.ascontiguousarray(variant)
        tree.set_engine("exact")
        w_off, w_idx = tree.query_radius_batch(variant, rr)
        tree.set_engine("auto")
        o, i = tree.query_radius_batch(variant, rr)
        assert np.array_equal(o, w_off) and np.array_equal(i, w_idx)
    tree.close()


def test_cosine_radius_through_shards(pn):
    """three virtual shards of a Cosine corpus, each large enough for its own tier: the sharded lists equal the single
    index's"""
    from petal_neighbors_amd.sharded import ShardedIndex
    n, dim, nq = 45_000, 48, 200
    pts = uniform((n, dim), 8401, np.float32) - np.float32(0.4)
    qs = uniform((nq, dim), 8402, np.float32) - np.float32(0.4)
    tree = pn.BallTree.new(pts, pn.distance.Cosine())
    _, d3 = tree.query_batch(qs, 3)
    r = np.float32(np.median(d3[:, 2]))
    want_off, want_idx = tree.query_radius_batch(qs, r)
    sh = ShardedIndex.from_host(pts, [0, 0, 0], metric=pn.distance.Cosine())
    off, idx = sh.query_radius_batch(qs, r)
    assert np.array_equal(np.asarray(off, dtype=np.uint64), want_off) and np.array_equal(np.asarray(idx, dtype=np.uint64), want_idx)
    sh.close()
    tree.close()
