"""Shard-local engine + HIP merge kernel on one GPU: several row shards of one corpus (what each
rank would hold), per-shard results with global indices (PN_OPT_INDEX_BASE), stacked the way an
all-gather leaves them, merged by pn_merge_topk_device_f32 -- must equal the single-index answer
bit for bit, for any shard count.  The distributed call pattern itself is covered
on CPU by test_sharded_gloo.py."""
import numpy as np
import pytest

from conftest import uniform

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shards", [1, 2, 3, 8])
@pytest.mark.parametrize("n,dim,nq,k", [(20000, 128, 200, 10), (1000, 16, 50, 40), (37, 3, 9, 5)])
def test_virtual_shards_merge_equals_single_index(pn, oracle_mod, shards, n, dim, nq, k):
    import torch
    from petal_neighbors_amd import _lib
    from petal_neighbors_amd.sharded import HipShardEngine, shard_bounds
    pts = uniform((n, dim), 77 + n, np.float32)
    if n > 100:
        pts[n - 1] = pts[0]  # duplicate across shards: tie broken by GLOBAL index
    qs = np.concatenate([pts[:3], uniform((nq - 3, dim), 78 + n, np.float32)])
    qd = torch.from_numpy(qs).to("cuda:0")
    k_part = min(k, max(shard_bounds(n, shards, 0)[1], 1))
    idx_parts = torch.full((shards, nq, k_part), -1, dtype=torch.int64, device="cuda:0")
    dst_parts = torch.full((shards, nq, k_part), float("nan"), dtype=torch.float32, device="cuda:0")
    eng = None
    for r in range(shards):
        lo, hi = shard_bounds(n, shards, r)
        if hi <= lo:
            continue
        eng = HipShardEngine(0)
        eng.build(pts[lo:hi], lo)
        li, ld = eng.query(qd, k_part)
        idx_parts[r, :, : li.shape[1]] = li
        dst_parts[r, :, : ld.shape[1]] = ld
    k_out = min(k, n)
    mi, md = eng.merge(idx_parts, dst_parts, k_out)
    torch.cuda.synchronize()
    want_i, want_d = oracle_mod.brute_knn(pts, qs, k)
    assert np.array_equal(mi.cpu().numpy().astype(np.uint64), want_i)
    assert md.cpu().numpy().tobytes() == want_d.tobytes()


@pytest.mark.parametrize("shards", [2, 8])
def test_packed_query_buffers_merge_in_place(pn, oracle_mod, shards):
    """The N > 1 fast path: every shard writes its top-k straight into the packed buffer an all-gather would send;
    the concatenation of those buffers, unpacked as strided views, merges to the single-index answer."""
    import torch
    from petal_neighbors_amd.sharded import HipShardEngine, shard_bounds
    n, dim, nq, k = 30000, 96, 333, 10
    pts = uniform((n, dim), 91, np.float32)
    qs = uniform((nq, dim), 92, np.float32)
    qd = torch.from_numpy(qs).to("cuda:0")
    bufs, eng = [], None
    for r in range(shards):
        lo, hi = shard_bounds(n, shards, r)
        eng = HipShardEngine(0)
        eng.build(pts[lo:hi], lo)
        bufs.append(eng.query_packed(qd, k))
    gathered = torch.cat(bufs)
    g_idx, g_dst = eng.unpack(gathered, shards, nq, k)
    mi, md = eng.merge(g_idx, g_dst, k)
    torch.cuda.synchronize()
    want_i, want_d = oracle_mod.brute_knn(pts, qs, k)
    assert np.array_equal(mi.cpu().numpy().astype(np.uint64), want_i)
    assert md.cpu().numpy().tobytes() == want_d.tobytes()


def test_sharded_ball_tree_world_size_one(pn, oracle_mod):
    """ShardedBallTree without a process group = one shard: the bench.py code path."""
    import torch
    pts = uniform((30000, 64), 5, np.float32)
    qs = uniform((300, 64), 6, np.float32)
    index = pn.ShardedBallTree(len(pts), lambda lo, hi: torch.from_numpy(pts[lo:hi]).to("cuda:0"))
    i, d = index.query_batch(torch.from_numpy(qs).to("cuda:0"), 10)
    torch.cuda.synchronize()
    want_i, want_d = oracle_mod.brute_knn(pts, qs, 10)
    assert np.array_equal(i.cpu().numpy().astype(np.uint64), want_i)
    assert d.cpu().numpy().tobytes() == want_d.tobytes()
    off, ids = index.query_radius_batch(torch.from_numpy(qs[:4]).to("cuda:0"), 2.5)
    for a in range(4):
        assert np.array_equal(ids[int(off[a]):int(off[a + 1])], oracle_mod.brute_radius(pts, qs[a], np.float32(2.5)))
