"""Row-sharded corpora through the C ABI (pn_sharded_*, csrc/sharded.hip) on one MI355X.

One GPU is what a test box has, so the shards are VIRTUAL: ``devices = [0] * G`` puts G row shards on GPU 0 -- the
same code as G GPUs except that the communicator has one rank: per-shard top-k with global indices straight into the
packed buffer, local merge, ONE ncclAllGather through RCCL (world size 1), final merge.  Results must equal the
oracle's brute force over the whole corpus bit for bit, for every shard count.  The one-process-per-GPU
entry (pn_sharded_create_rank_device_f32, what bench.py launches) runs here at world size 1, in-process and under a
real ``torch.distributed`` nccl group in a child process.
"""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from conftest import ROOT, uniform

pytestmark = pytest.mark.gpu


def _same(ai, ad, bi, bd):
    return np.array_equal(np.asarray(ai).astype(np.uint64), np.asarray(bi).astype(np.uint64)) and \
        np.ascontiguousarray(ad).tobytes() == np.ascontiguousarray(bd).tobytes()


@pytest.mark.parametrize("shards", [1, 2, 3, 8])
@pytest.mark.parametrize("n,dim,nq,k", [(20000, 128, 200, 10), (1000, 16, 50, 40), (37, 3, 9, 5), (5, 4, 7, 3)])
def test_virtual_shards_through_the_abi(pn, oracle_mod, shards, n, dim, nq, k):
    from petal_neighbors_amd import _lib
    pts = uniform((n, dim), 177 + n, np.float32)
    if n > 100:
        pts[n - 1] = pts[0]  # duplicate across shards: the tie is broken by the GLOBAL index
    h = min(3, n)
    qs = np.concatenate([pts[:h], uniform((nq - h, dim), 178 + n, np.float32)])
    sh = pn.ShardedIndex.from_host(pts, [0] * shards)
    assert (sh.n, sh.dim, sh.n_shards, sh.world, sh.local_rows) == (n, dim, shards, 1, n)
    sh.set_option(_lib.PN_OPT_EXCHANGE_ALWAYS, 1)
    want_i, want_d = oracle_mod.brute_knn(pts, qs, k)
    gi, gd = sh.query_batch(qs, k)
    assert gi.shape == (nq, min(k, n))
    assert _same(gi, gd, want_i, want_d)
    # k beyond a shard's rows (absent slots in the packed buffers), and beyond the corpus
    for kk in (n // max(shards, 1) + 3, n + 5):
        if n > 5000:
            continue  # (large k on the large corpus: test_large_k_is_not_bounded_by_the_merge_kernels_lds)
        wi, wd = oracle_mod.brute_knn(pts, qs, kk)
        gi, gd = sh.query_batch(qs, kk)
        assert _same(gi, gd, wi, wd), kk
    assert sh.query_batch(qs, 0)[0].shape == (nq, 0)
    # radius: ascending global rows, shard after shard
    _, d = oracle_mod.brute_knn(pts, qs, min(4, n))
    for r in (float(np.median(d[:, -1])), 0.0, 1e9):
        off, ids = sh.query_radius_batch(qs, r)
        assert off[0] == 0 and off[-1] == len(ids)
        for a in range(nq):
            assert np.array_equal(ids[int(off[a]):int(off[a + 1])], oracle_mod.brute_radius(pts, qs[a], np.float32(r))), (r, a)
    sh.close()


def test_abi_errors_and_single_shard_shortcut(pn, oracle_mod):
    from petal_neighbors_amd import _lib
    with pytest.raises(pn.ArrayError.Empty):
        pn.ShardedIndex.from_host(np.zeros((0, 4), dtype=np.float32), [0, 0])
    with pytest.raises(pn.ArrayError.NotContiguous):
        pn.ShardedIndex.from_host(np.asfortranarray(uniform((6, 4), 1)), [0, 0])
    with pytest.raises(pn.PetalError):
        pn.ShardedIndex.from_host(uniform((6, 4), 1), [0, 99])
    pts, qs = uniform((9000, 32), 3), uniform((300, 32), 4)
    want = oracle_mod.brute_knn(pts, qs, 7)
    sh = pn.ShardedIndex.from_host(pts, [0])
    a = sh.query_batch(qs, 7)  # one shard: straight into the caller's buffers
    sh.set_option(_lib.PN_OPT_EXCHANGE_ALWAYS, 1)
    b = sh.query_batch(qs, 7)  # the same through pack -> RCCL all-gather (world size 1) -> merge
    assert _same(*a, *want) and _same(*b, *want)
    with pytest.raises(pn.PetalError):
        sh.set_option(_lib.PN_OPT_INDEX_BASE, 5)
    for eng in ("exact", "mfma", "bf16", "auto"):
        sh.set_engine(eng)
        assert _same(*sh.query_batch(qs, 7), *want), eng
    st = sh.stats()
    assert st["queries"] == 6 * 300
    sh.close()


def test_device_queries_and_chunk_overlap(pn, oracle_mod):
    """pn_sharded_query_device_f32 with more than 131 072 queries: chunks whose exchange + merge run on the second
    stream while the next chunk is filtered (two buffer sets).  3 virtual shards; against the unsharded index on
    every query and the oracle on a sample."""
    import torch
    from petal_neighbors_amd import _lib
    n, dim, nq, k = 30000, 16, 300_000, 5
    pts = uniform((n, dim), 61)
    qs = uniform((nq, dim), 62)
    qd = torch.from_numpy(qs).to("cuda:0")
    sh = pn.ShardedIndex.from_host(pts, [0, 0, 0])
    i1, d1 = sh.query_device(qd, k)
    i2, d2 = sh.query_device(qd, k)  # back to back on one stream: buffer reuse across calls
    tree = pn.BallTree.euclidean(pts)
    ti, td = tree.query_device(qd, k)
    torch.cuda.synchronize()
    assert torch.equal(i1, ti) and torch.equal(d1.view(torch.int32), td.view(torch.int32))
    assert torch.equal(i2, ti) and torch.equal(d2.view(torch.int32), td.view(torch.int32))
    sel = np.linspace(0, nq - 1, 64).astype(np.int64)
    wi, wd = oracle_mod.brute_knn(pts, qs[sel], k)
    assert _same(i1.cpu().numpy()[sel], d1.cpu().numpy()[sel], wi, wd)
    sh.close()


def test_two_streams_share_one_sharded_handle(pn, oracle_mod):
    """pn_sharded_query_device_f32 is asynchronous: two calls on ONE handle from two streams must not run side by side
    on the handle's pack / gathered / local-parts buffers.  The handle orders a call behind the previous call's
    end-of-use event when the stream changes (csrc/sharded.hip, acquire_dev).  3 virtual shards, different query sets
    and different k per stream (different buffer shapes), many alternations, every answer against the oracle."""
    import torch
    n, dim = 30000, 32
    pts = uniform((n, dim), 65)
    sh = pn.ShardedIndex.from_host(pts, [0, 0, 0])
    sets = [(uniform((3000, dim), 66), 10), (uniform((1700, dim), 67), 23)]
    want = [oracle_mod.brute_knn(pts, q, k) for q, k in sets]
    qd = [torch.from_numpy(q).to("cuda:0") for q, _ in sets]
    streams = [torch.cuda.Stream(device="cuda:0"), torch.cuda.Stream(device="cuda:0")]
    torch.cuda.synchronize()
    outs = []
    for rep in range(6):
        for t in (0, 1):
            with torch.cuda.stream(streams[t]):
                outs.append((t, *sh.query_device(qd[t], sets[t][1])))  # returns at once; nothing waits in between
    torch.cuda.synchronize()
    for t, i, d in outs:
        assert _same(i.cpu().numpy(), d.cpu().numpy(), *want[t]), t
    sh.close()


@pytest.mark.parametrize("shards", [2, 8])
def test_large_k_is_not_bounded_by_the_merge_kernels_lds(pn, oracle_mod, shards):
    """BallTree::query has no limit on k (src/ball_tree.rs:102-121).  shards x k x 12 bytes beyond 64 KiB used to be
    refused by the LDS-resident merge (8 shards: k <= 682); now the merge ranks by binary search over the sorted parts
    (select.hip, merge_sorted_topk_kernel).  Local merge of the virtual shards AND the merge behind the all-gather."""
    from petal_neighbors_amd import _lib
    n, dim, nq = 24000, 8, 40
    pts = uniform((n, dim), 68)
    pts[n - 1] = pts[0]
    pts[7] = pts[n // 2 + 1]  # equal distances across shards: the global index decides
    qs = np.concatenate([pts[:2], uniform((nq - 2, dim), 69)])
    sh = pn.ShardedIndex.from_host(pts, [0] * shards)
    sh.set_option(_lib.PN_OPT_EXCHANGE_ALWAYS, 1)
    for k in (1500, n // shards + 7, 6000):
        wi, wd = oracle_mod.brute_knn(pts, qs, k)
        gi, gd = sh.query_batch(qs, k)
        assert _same(gi, gd, wi, wd), (shards, k)
    sh.close()


def test_rank_entry_at_world_size_one(pn, oracle_mod):
    """One process per GPU, world size 1, no torch.distributed: communicator id from pn_comm_unique_id,
    ncclCommInitRank, exchange forced through RCCL."""
    import torch
    from petal_neighbors_amd import _lib
    from petal_neighbors_amd.sharded import ShardedBallTree
    n, dim, nq, k = 12000, 64, 150, 10
    pts, qs = uniform((n, dim), 71), uniform((nq, dim), 72)
    torch.cuda.set_device(0)
    index = ShardedBallTree(n, lambda lo, hi: torch.from_numpy(pts[lo:hi]).to("cuda:0"))
    assert (index.world, index.lo, index.hi) == (1, 0, n)
    index.engine.tree.set_option(_lib.PN_OPT_EXCHANGE_ALWAYS, 1)
    gi, gd = index.query_batch(torch.from_numpy(qs).to("cuda:0"), k)
    torch.cuda.synchronize()
    assert _same(gi.cpu().numpy(), gd.cpu().numpy(), *oracle_mod.brute_knn(pts, qs, k))
    off, ids = index.query_radius_batch(torch.from_numpy(qs[:20]).to("cuda:0"), 2.4)
    for a in range(20):
        assert np.array_equal(ids[int(off[a]):int(off[a + 1])], oracle_mod.brute_radius(pts, qs[a], np.float32(2.4)))


_NCCL_CHILD = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from conftest import uniform
import oracle
import petal_neighbors_amd as pn
from petal_neighbors_amd import _lib
from petal_neighbors_amd.sharded import ShardedBallTree, HipShardEngine, AbiShardEngine
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
n, dim, nq, k = 15000, 96, 120, 10
pts, qs = uniform((n, dim), 81), uniform((nq, dim), 82)
want_i, want_d = oracle.brute_knn(pts, qs, k)
qd = torch.from_numpy(qs).to("cuda:0")
for name, eng in (("abi", AbiShardEngine(0)), ("torch", HipShardEngine(0))):
    index = ShardedBallTree(n, lambda lo, hi: torch.from_numpy(pts[lo:hi]).to("cuda:0"), engine=eng)
    if name == "abi":
        index.engine.tree.set_option(_lib.PN_OPT_EXCHANGE_ALWAYS, 1)
    gi, gd = index.query_batch(qd, k)
    torch.cuda.synchronize()
    assert np.array_equal(gi.cpu().numpy().astype(np.uint64), want_i), name
    assert gd.cpu().numpy().tobytes() == want_d.tobytes(), name
# the torch-side exchange on HIP tensors: one all_gather_into_tensor of the packed buffer at world size 1
eng = HipShardEngine(0)
eng.build(torch.from_numpy(pts).to("cuda:0"), 0)
mine = eng.query_packed(qd, k)
gathered = torch.empty_like(mine)
dist.all_gather_into_tensor(gathered, mine)
gi, gd = eng.merge(*eng.unpack(gathered, 1, nq, k), k)
torch.cuda.synchronize()
assert np.array_equal(gi.cpu().numpy().astype(np.uint64), want_i) and gd.cpu().numpy().tobytes() == want_d.tobytes()
dist.barrier()
dist.destroy_process_group()
print("NCCL_WORLD1_OK")
"""


def test_nccl_backend_at_world_size_one(tmp_path):
    """torch.distributed's nccl (= RCCL) backend on the test GPU at world size 1: both exchange
    paths -- behind the ABI and torch's all_gather_into_tensor -- in a child process (a process group is global
    state; the child is started, never exec'ed into)."""
    script = tmp_path / "nccl_child.py"
    script.write_text(_NCCL_CHILD.format(root=ROOT))
    r = None
    for port in (29541, 29547, 29553):  # a busy port only costs a retry
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
        if r.returncode == 0 or "EADDRINUSE" not in r.stderr:
            break
    assert r.returncode == 0 and "NCCL_WORLD1_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_two_host_threads_share_one_handle(pn, oracle_mod):
    """`Euclidean: Sync` (src/distance.rs:19): one tree queried from two threads at once -- every call works in its
    own pooled workspace.  Host API and device API, k-NN and radius, against the oracle."""
    import torch
    n, dim, k = 40000, 128, 10
    pts = uniform((n, dim), 91)
    tree = pn.BallTree.euclidean(pts)
    sets = [uniform((256, dim), 92 + t) for t in range(4)]
    want = [oracle_mod.brute_knn(pts, q, k) for q in sets]
    got, errs = [None] * 4, []

    def host(t):
        try:
            for _ in range(6):
                got[t] = tree.query_batch(sets[t], k)
                assert _same(*got[t], *want[t])
            off, ids = tree.query_radius_batch(sets[t][:40], 3.3)
            for a in range(40):
                assert np.array_equal(ids[int(off[a]):int(off[a + 1])], oracle_mod.brute_radius(pts, sets[t][a], np.float32(3.3)))
        except Exception as e:  # noqa: BLE001
            errs.append((t, repr(e)))

    def device(t):
        try:
            st = torch.cuda.Stream(device="cuda:0")
            qd = torch.from_numpy(sets[t]).to("cuda:0")
            torch.cuda.synchronize()
            for _ in range(6):
                with torch.cuda.stream(st):
                    i, d = tree.query_device(qd, k)
                st.synchronize()
                assert _same(i.cpu().numpy(), d.cpu().numpy(), *want[t])
        except Exception as e:  # noqa: BLE001
            errs.append((t, repr(e)))

    ths = [threading.Thread(target=host, args=(0,)), threading.Thread(target=host, args=(1,)),
           threading.Thread(target=device, args=(2,)), threading.Thread(target=device, args=(3,))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    assert tree.stats()["queries"] == 4 * 6 * 256


def _f64(shape, seed):
    rng = np.random.default_rng(seed)
    return rng.random(shape)  # 53 random bits: values no f32 represents


@pytest.mark.parametrize("shards", [1, 3, 8])
@pytest.mark.parametrize("n,dim,nq,k", [(20000, 64, 150, 10), (900, 16, 40, 35), (37, 3, 9, 5)])
def test_f64_virtual_shards_through_the_abi(pn, oracle_mod, shards, n, dim, nq, k):
    """The sharded handle is generic over the element type like BallTree<A, M> (src/ball_tree.rs:26-30): an f64 corpus
    gives f64 shards (bf16 filter, f64 re-rank), distances travel and merge as f64 -- results equal the oracle's f64
    brute force bit for bit, k-NN and radius, host and device entry points."""
    import torch
    from petal_neighbors_amd import _lib
    pts = _f64((n, dim), 277 + n)
    if n > 100:
        pts[n - 1] = pts[0]  # duplicate across shards: the tie is broken by the GLOBAL index
    h = min(3, n)
    qs = np.concatenate([pts[:h], _f64((nq - h, dim), 278 + n)])
    sh = pn.ShardedIndex.from_host(pts, [0] * shards)
    assert sh.dtype == np.float64 and (sh.n, sh.dim, sh.n_shards, sh.world) == (n, dim, shards, 1)
    sh.set_option(_lib.PN_OPT_EXCHANGE_ALWAYS, 1)
    want_i, want_d = oracle_mod.brute_knn(pts, qs, k)
    assert want_d.dtype == np.float64
    gi, gd = sh.query_batch(qs, k)
    assert gd.dtype == np.float64 and _same(gi, gd, want_i, want_d)
    if n <= 5000:  # k beyond a shard's rows (absent slots in the packed buffers), and beyond the corpus
        for kk in (n // shards + 3, n + 5):
            wi, wd = oracle_mod.brute_knn(pts, qs, kk)
            gi, gd = sh.query_batch(qs, kk)
            assert _same(gi, gd, wi, wd), kk
    # device entry point (one process drives one GPU)
    qd = torch.from_numpy(qs).to("cuda:0")
    di, dd = sh.query_device(qd, k)
    torch.cuda.synchronize()
    assert dd.dtype == torch.float64 and _same(di.cpu().numpy(), dd.cpu().numpy(), want_i, want_d)
    with pytest.raises(Exception):
        sh.query_device(qd.float(), k)   # an f32 batch on an f64 handle
    # radius
    _, d = oracle_mod.brute_knn(pts, qs, min(4, n))
    r = float(np.median(d[:, -1]))
    off, ids = sh.query_radius_batch(qs, r)
    assert off[0] == 0 and off[-1] == len(ids)
    for a in range(nq):
        assert np.array_equal(ids[int(off[a]):int(off[a + 1])], oracle_mod.brute_radius(pts, qs[a], np.float64(r))), a
    sh.close()


def test_f64_rank_entry_at_world_size_one(pn, oracle_mod):
    """pn_sharded_create_rank_device_f64: one process per GPU at world size 1, f64 rows already in HBM, exchange forced."""
    import torch
    from petal_neighbors_amd import _lib
    n, dim, nq, k = 9000, 48, 100, 8
    pts, qs = _f64((n, dim), 91), _f64((nq, dim), 92)
    rows = torch.from_numpy(pts).to("cuda:0")
    sh = pn.ShardedIndex.from_rank_device(rows, n, 0, 1, pn.ShardedIndex.unique_id(), 0)
    assert sh.dtype == np.float64
    sh.set_option(_lib.PN_OPT_EXCHANGE_ALWAYS, 1)
    gi, gd = sh.query_device(torch.from_numpy(qs).to("cuda:0"), k)
    torch.cuda.synchronize()
    assert _same(gi.cpu().numpy(), gd.cpu().numpy(), *oracle_mod.brute_knn(pts, qs, k))
    sh.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shards", [2, 5])
def test_cosine_shards_equal_the_single_cosine_index(pn, dtype, shards):
    """BallTree::new(points, Cosine) over row shards: every shard a Cosine index, the part merge on the order-preserving
    keys of ALL floats (rows parallel to a query have distances a few ulp around zero, some below it).  Answers equal the
    single Cosine index's bit for bit, k-NN and radius."""
    from petal_neighbors_amd import _lib
    rng = np.random.default_rng(5)
    n, dim, nq, k = 6000, 24, 60, 9
    pts = (rng.random((n, dim)) - 0.3).astype(dtype)
    pts[100:140] = pts[7] * np.linspace(0.5, 3.0, 40, dtype=dtype)[:, None]   # parallel to row 7: cosine distance ~ 0
    pts[5000:5020] = pts[7] * dtype(1.7)
    qs = np.concatenate([pts[7:8], pts[3:4] * dtype(2.5), (rng.random((nq - 2, dim)) - 0.3).astype(dtype)])
    from petal_neighbors_amd.distance import Cosine
    one = pn.BallTree.new(pts, Cosine())
    sh = pn.ShardedIndex.from_host(pts, [0] * shards, metric=Cosine())
    sh.set_option(_lib.PN_OPT_EXCHANGE_ALWAYS, 1)
    wi, wd = one.query_batch(qs, k)
    gi, gd = sh.query_batch(qs, k)
    assert gd.dtype == np.dtype(dtype) and _same(gi, gd, wi, wd)
    assert (wd[0] <= 0).any() or (wd[:2] < 1e-6).all()     # the near-zero group is really in play
    r = float(np.median(wd[:, 4]))
    o1, i1 = one.query_radius_batch(qs, r)
    o2, i2 = sh.query_radius_batch(qs, r)
    assert np.array_equal(o1, o2) and np.array_equal(i1, i2)
    sh.close()
