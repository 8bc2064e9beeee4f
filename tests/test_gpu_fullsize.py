"""Parity of the BENCHED engine at the BENCHED sizes (need an MI355X).

BASELINE.json configs[1..4] at full scale, through the C ABI, with the engine bench.py times
(``auto`` = bf16 filter first).  No full oracle run fits at these sizes, so every config is checked by
  * size-independent properties on EVERY query: ascending distances, no duplicate neighbours,
    indices in range, and -- on a sample of (query, neighbour) pairs -- the returned distance equal to the
    scalar metric recomputed for that index (bit-exact);
  * bit-equality with the exact VALU engine (bit-exact by construction, parity-tested against the oracle
    at small sizes) on a few thousand queries -- all of them at configs[1];
  * bit-equality with the CPU oracle's brute force on a sample of queries (the corpus is streamed to the
    host in chunks; the per-chunk top-k are merged under the (distance, index) order).
The pattern is the reference's own property test, src/ball_tree.rs:742-765 (tree == naive scan).
Scale-only code covered here: 12 segments per query tile + shared scout (configs[1]); grid-in-rounds plan with
100 000 queries, k = 100 (configs[2]); corpora above 2^32 elements (configs[3]: 7.7e9); the > 262 144-query
chunk loop of pn_query_device_f32 (configs[4] shard, 300 000 queries).
"""
import threading

import numpy as np
import pytest

from conftest import uniform

pytestmark = pytest.mark.gpu

SEED_P, SEED_Q = 0x5EED0001, 0x5EED0002


def _gen(n, dim, seed):
    import torch
    from petal_neighbors_amd import _lib
    t = torch.empty((n, dim), dtype=torch.float32, device="cuda:0")
    assert _lib.lib().pn_fill_uniform_device_f32(t.data_ptr(), n * dim, seed, 0, 0, None) == 0, _lib.last_error()
    torch.cuda.synchronize()
    return t


def _oracle_knn_streamed(oracle_mod, pts_t, qs, k, chunk_rows=2_000_000, threads=12):
    """oracle.brute_knn over a device-resident corpus: rows come to the host chunk by chunk, each query's per-chunk
    top-k (global indices) are merged under (distance, index).  Queries are spread over threads (ctypes drops the GIL)."""
    n = pts_t.shape[0]
    nq = qs.shape[0]
    best_i = [np.empty(0, dtype=np.uint64) for _ in range(nq)]
    best_d = [np.empty(0, dtype=np.float32) for _ in range(nq)]
    for lo in range(0, n, chunk_rows):
        hi = min(n, lo + chunk_rows)
        ph = pts_t[lo:hi].cpu().numpy()
        out = [None] * nq

        def work(t):
            for a in range(t, nq, threads):
                out[a] = oracle_mod.brute_knn(ph, qs[a:a + 1], k)
        ts = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for a in range(nq):
            i, d = out[a]
            ci = np.concatenate([best_i[a], i[0] + np.uint64(lo)])
            cd = np.concatenate([best_d[a], d[0]])
            # distances here are finite and >= 0: their bit patterns order like the values
            order = np.lexsort((ci, cd.view(np.uint32)))[:k]
            best_i[a], best_d[a] = ci[order], cd[order]
        del ph
    return np.stack(best_i), np.stack(best_d)


def _properties(idx_t, dist_t, n):
    """ascending, in range, no duplicates -- on every query, on the device"""
    import torch
    assert bool((dist_t[:, 1:] >= dist_t[:, :-1]).all()), "distances not ascending"
    assert bool(((idx_t >= 0) & (idx_t < n)).all()), "index out of range"
    s = torch.sort(idx_t, dim=1).values
    assert bool((s[:, 1:] != s[:, :-1]).all()), "duplicate neighbour"
    assert bool(torch.isfinite(dist_t).all())


def _metric_rederived(pn, pts_t, qs_t, idx_t, dist_t, n_queries=48):
    """returned distance == Euclidean::distance(query, points[idx]) bit for bit, on sampled queries"""
    import torch
    nq, k = idx_t.shape
    sel = torch.linspace(0, nq - 1, n_queries, device="cuda:0").long()
    qi = idx_t[sel]
    rows = pts_t[qi.reshape(-1)].cpu().numpy().reshape(n_queries, k, -1)
    qh = qs_t[sel].cpu().numpy()
    dh = dist_t[sel].cpu().numpy()
    m = pn.distance.Euclidean()
    for a in range(n_queries):
        for j in range(0, k, max(1, k // 10)):
            assert m.distance(qh[a], rows[a, j]).tobytes() == dh[a, j].tobytes(), (a, j)


def _same(ai, ad, bi, bd):
    return np.array_equal(np.asarray(ai).astype(np.uint64), np.asarray(bi).astype(np.uint64)) and \
        np.ascontiguousarray(ad).tobytes() == np.ascontiguousarray(bd).tobytes()


def _run(pn, oracle_mod, n, dim, nq, k, n_exact, n_oracle, max_fallback, chunk_rows=2_000_000):
    """the common body: auto engine on all nq queries; exact engine on n_exact of them (spread over the whole
    range so that every query chunk is sampled); oracle on n_oracle of them"""
    import torch
    pts_t = _gen(n, dim, SEED_P)
    qs_t = _gen(nq, dim, SEED_Q)
    tree = pn.BallTree.from_device(pts_t)
    assert tree.bf16_eligible
    tree.set_engine("auto")
    idx_t, dist_t = tree.query_device(qs_t, k)
    torch.cuda.synchronize()
    st = tree.stats(reset=True)
    assert st["queries"] == nq
    assert st["fallback_queries"] <= max_fallback, st
    _properties(idx_t, dist_t, n)
    _metric_rederived(pn, pts_t, qs_t, idx_t, dist_t)
    # exact engine (bit-exact by construction) on a spread sample
    sel = torch.linspace(0, nq - 1, n_exact, device="cuda:0").long() if n_exact < nq else torch.arange(nq, device="cuda:0")
    tree.set_engine("exact")
    ei, ed = tree.query_device(qs_t[sel].contiguous(), k)
    torch.cuda.synchronize()
    assert torch.equal(ei, idx_t[sel]), "auto and exact engines disagree on indices"
    assert torch.equal(ed.view(torch.int32), dist_t[sel].view(torch.int32)), "auto and exact engines disagree on distances"
    # CPU oracle on a smaller spread sample
    osel = torch.linspace(0, nq - 1, n_oracle, device="cuda:0").long()
    oi, od = _oracle_knn_streamed(oracle_mod, pts_t, qs_t[osel].cpu().numpy(), k, chunk_rows)
    assert _same(idx_t[osel].cpu().numpy(), dist_t[osel].cpu().numpy(), oi, od), "GPU result differs from the oracle"
    return tree, pts_t, qs_t, idx_t, dist_t


def _cleanup(*objs):
    import gc
    import torch
    for o in objs:
        if hasattr(o, "close"):
            o.close()
    del objs
    gc.collect()
    torch.cuda.empty_cache()


def test_c2_headline_auto_engine_all_queries(pn, oracle_mod):
    """BASELINE.json configs[1], exactly what bench.py times: 1M x 128 f32, ALL 10 000 queries, k = 10, engine auto
    (bf16 filter, 12 segments per query tile, shared scout).  Exact engine on all 10 000; oracle on 256."""
    import torch
    n, dim, nq, k = 1_000_000, 128, 10_000, 10
    tree, pts_t, qs_t, idx_t, dist_t = _run(pn, oracle_mod, n, dim, nq, k, n_exact=nq, n_oracle=256, max_fallback=8,
                                            chunk_rows=1_000_000)
    # device generator == oracle generator, bit for bit
    head = oracle_mod.fill_uniform(1000 * dim, SEED_P).reshape(1000, dim)
    assert pts_t[:1000].cpu().numpy().tobytes() == head.tobytes()
    # the f32 MFMA tier agrees as well, and the host API returns the same as the device API
    if tree.mfma_eligible:
        tree.set_engine("mfma")
        mi, md = tree.query_device(qs_t, k)
        torch.cuda.synchronize()
        assert torch.equal(mi, idx_t) and torch.equal(md.view(torch.int32), dist_t.view(torch.int32))
    tree.set_engine("auto")
    hi, hd = tree.query_batch(qs_t[:512].cpu().numpy(), k)
    assert _same(hi, hd, idx_t[:512].cpu().numpy(), dist_t[:512].cpu().numpy())
    _cleanup(tree)


def test_c3_ten_million_rows_k100_and_radius(pn, oracle_mod):
    """configs[2]: 10M x 128 f32, 100 000 queries, k = 100 (grid-in-rounds plan, 41-slot buffers) and
    query_radius r = 0.5.  Exact engine on 2 048 queries, oracle on 12."""
    import torch
    n, dim, nq, k = 10_000_000, 128, 100_000, 100
    tree, pts_t, qs_t, idx_t, dist_t = _run(pn, oracle_mod, n, dim, nq, k, n_exact=2048, n_oracle=12,
                                            max_fallback=nq // 1000)
    # radius leg.  Independent uniform queries have no row within 0.5 (nearest ~3.4): the config measures the scan.
    # So the batch also carries corpus rows (answer contains the row itself) and a second radius near the
    # 3rd-neighbour distance where lists are non-empty; engine auto vs exact engine, and the oracle on a few.
    tree.set_engine("auto")
    qh = qs_t[:20_000].cpu().numpy()
    rows = [0, 1, 4_999_999, 9_999_999]
    qh[:4] = pts_t[torch.tensor(rows, device="cuda:0")].cpu().numpy()
    off, ids = tree.query_radius_batch(qh, np.float32(0.5))
    assert int(off[-1]) == 4 and ids.tolist() == rows, (off[:6], ids[:8])
    # (distances concentrate at this scale: the 1st and the 100th neighbour are a few percent apart, so the radius
    # is taken at the median NEAREST-neighbour distance and lists hold from zero to a few dozen rows)
    r1 = np.float32(dist_t[:20_000, 0].median().item())
    off, ids = tree.query_radius_batch(qh[:4096], r1)
    assert int(off[-1]) >= 2048
    tree.set_engine("exact")
    xoff, xids = tree.query_radius_batch(qh[:512], r1)
    assert np.array_equal(off[:513], xoff) and np.array_equal(ids[: int(off[512])], xids)
    # the k-NN answer determines the radius answer wherever the k-th neighbour lies outside the radius
    dh, ih = dist_t[:256].cpu().numpy(), idx_t[:256].cpu().numpy()
    checked = 0
    for a in range(4, 256):
        if dh[a, -1] < r1:
            continue
        want = np.sort(ih[a][dh[a] < r1]).astype(np.uint64)
        assert np.array_equal(ids[int(off[a]):int(off[a + 1])], want), a
        checked += 1
    assert checked >= 200
    # and the oracle's brute force over the whole corpus for three queries (the first is a corpus row)
    ph = pts_t.cpu().numpy()
    for a in (0, 5, 300):
        assert np.array_equal(ids[int(off[a]):int(off[a + 1])], oracle_mod.brute_radius(ph, qh[a], r1)), a
    del ph
    _cleanup(tree)


def test_c4_wide_rows_beyond_2_pow_32_elements(pn, oracle_mod):
    """configs[3] on one GPU: 10M x 768 f32 = 7.68e9 elements (> 2^32), 100 000 queries, k = 10 (K-chunked bf16
    kernel, grid in rounds).  Exact engine on 1 024 queries, oracle on 12."""
    n, dim, nq, k = 10_000_000, 768, 100_000, 10
    tree, *_ = _run(pn, oracle_mod, n, dim, nq, k, n_exact=1024, n_oracle=12, max_fallback=nq // 500,
                    chunk_rows=1_000_000)
    _cleanup(tree)


def test_c5_shard_query_chunk_loop(pn, oracle_mod):
    """One of the 8 row shards of configs[4]: 12.5M x 96 f32, 300 000 queries (more than one 262 144-query chunk of
    pn_query_device_f32), k = 10.  Exact engine on 2 048 queries spread over both chunks, oracle on 12."""
    n, dim, nq, k = 12_500_000, 96, 300_000, 10
    tree, *_ = _run(pn, oracle_mod, n, dim, nq, k, n_exact=2048, n_oracle=12, max_fallback=nq // 1000)
    _cleanup(tree)


def test_c5_whole_on_one_gpu(pn, oracle_mod):
    """configs[4] WHOLE on one GPU (the one config that had only been run as a shard under the test run):
    10^8 x 96 f32 (38.4 GB of rows + 20.8 GB of tile images) and 10^6 queries, k = 10 -- four 262 144-query chunks over
    1.56 M row tiles.  Properties on all 10^6 answers, the exact engine on 1 024 of them, the oracle's brute force over
    the whole corpus (streamed to the host in chunks) on 8."""
    n, dim, nq, k = 100_000_000, 96, 1_000_000, 10
    tree, *_ = _run(pn, oracle_mod, n, dim, nq, k, n_exact=1024, n_oracle=8, max_fallback=nq // 2000,
                    chunk_rows=2_500_000)
    _cleanup(tree)


@pytest.mark.parametrize("dtype,n,dim", [(np.float32, 16384, 128), (np.float64, 16384, 16), (np.float32, 8192, 768)])
def test_pairwise_at_the_benched_sizes(pn, oracle_mod, dtype, n, dim):
    """distance::pairwise at the sizes of the roofline table (tools/bench_pairwise.py): the HBM-resident
    entry, 33 000 tiles of the upper triangle.  Checked by properties on the whole matrix (symmetric bit for bit, zero
    diagonal, no NaN) and against the scalar metric (the oracle's bits) on sampled pairs spread over all tile rows."""
    import torch
    x = uniform((n, dim), 8800 + dim, dtype)
    xd = torch.from_numpy(x).to("cuda:0")
    got = pn.distance.pairwise_device(xd)
    torch.cuda.synchronize()
    view = torch.int32 if dtype == np.float32 else torch.int64
    assert torch.equal(got.view(view), got.t().contiguous().view(view))
    assert bool((torch.diagonal(got) == 0).all()) and not bool(torch.isnan(got).any())
    rng = np.random.default_rng(dim)
    ii = np.concatenate([rng.integers(0, n, 3000), np.arange(0, n, 257), [0, n - 1, n - 1]])
    jj = np.concatenate([rng.integers(0, n, 3000), np.arange(0, n, 257)[::-1], [n - 1, 0, n - 2]])
    vals = got[torch.from_numpy(ii).to("cuda:0"), torch.from_numpy(jj).to("cuda:0")].cpu().numpy()
    for a in range(len(ii)):
        want = oracle_mod.euclidean(x[ii[a]], x[jj[a]]) if ii[a] != jj[a] else dtype(0)
        assert vals[a].tobytes() == dtype(want).tobytes(), (ii[a], jj[a])
