"""pn_query_radius_device_{f32,f64} (round 4): BallTree::query_radius (src/ball_tree.rs:137-142, 250-294) with queries and
results in HBM and no host round trip -- counts, exclusive scan and fill on the device, caller-supplied capacity, the
total left in HBM.  Same lists as the host entry point (strict '<', ascending index), which is itself checked against
the oracle's brute force in test_gpu_parity.py / test_gpu_bf16.py."""
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


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,dim,nq", [(30000, 32, 700), (5000, 128, 33), (300, 5, 70), (70000, 96, 1500)])
def test_radius_device_equals_the_host_entry_and_the_oracle(pn, oracle_mod, dtype, n, dim, nq):
    pts = uniform((n, dim), 5100 + dim, dtype)
    qs = np.concatenate([pts[:3], uniform((nq - 3, dim), 5200 + dim, dtype)])   # three corpus rows: non-empty lists
    tree = pn.BallTree.euclidean(pts)
    _, d3 = oracle_mod.brute_knn(pts, qs[:64], 3)
    r = dtype(np.median(d3[:, 2]))
    want_off, want_idx = tree.query_radius_batch(qs, r)
    total = int(want_off[-1])
    assert total >= 3
    off, idx, tot = _device_lists(tree, qs, r, total + 7)
    assert tot == total and np.array_equal(off, want_off) and np.array_equal(idx[:total], want_idx)
    for a in (0, 1, 2, nq - 1):  # and the oracle itself on a few
        assert np.array_equal(idx[int(off[a]):int(off[a + 1])], oracle_mod.brute_radius(pts, qs[a], r)), a
    # a buffer that is too small: offsets and total are complete, the first `capacity` entries are in place
    cap = max(total // 2, 1)
    off2, idx2, tot2 = _device_lists(tree, qs, r, cap)
    assert tot2 == total and np.array_equal(off2, want_off) and np.array_equal(idx2[:cap], want_idx[:cap])
    # counting only
    import torch
    qd = torch.from_numpy(qs).to("cuda:0")
    offs0, _, tot0 = tree.query_radius_device(qd, r, 0)
    torch.cuda.synchronize()
    assert int(tot0.item()) == total and np.array_equal(offs0.cpu().numpy().astype(np.uint64), want_off)


def test_radius_device_dense_neighbourhoods_take_the_listed_exact_pass(pn, oracle_mod):
    """queries inside a clump overflow the filter's survivor lists: they are LISTED on the device and answered by the
    exact two-pass scan through that list; a NaN query and a zero radius come back empty; engines agree"""
    rng = np.random.default_rng(61)
    base = uniform((200_000, 16), 6101)
    clump = (base[777] + 0.002 * rng.standard_normal((3000, 16))).astype(np.float32)
    pts = np.concatenate([base, clump]).astype(np.float32)
    qs = np.concatenate([uniform((200, 16), 6102), clump[:3] + np.float32(0.0005)]).astype(np.float32)
    qs[5, 2] = np.nan
    tree = pn.BallTree.euclidean(pts)
    _, d = oracle_mod.brute_knn(pts, qs[:30], 3)
    r = np.float32(np.median(d[~np.isnan(d[:, 2]), 2]))
    want_off, want_idx = tree.query_radius_batch(qs, r)
    total = int(want_off[-1])
    assert total > 3000                                   # the clump queries hold thousands of rows
    off, idx, tot = _device_lists(tree, qs, r, total)
    assert tot == total and np.array_equal(off, want_off) and np.array_equal(idx[:total], want_idx)
    assert off[6] == off[5]                               # the NaN query's list is empty
    for eng in ("exact", "bf16"):
        tree.set_engine(eng)
        off_e, idx_e, tot_e = _device_lists(tree, qs, r, total)
        assert tot_e == total and np.array_equal(off_e, want_off) and np.array_equal(idx_e[:total], want_idx), eng
    tree.set_engine("auto")
    off0, _, tot0 = _device_lists(tree, qs, np.float32(0.0), 8)
    assert tot0 == 0 and not off0.any()


def test_radius_device_cosine_index_and_async(pn, oracle_mod):
    """a Cosine index (round 4: the bf16 tier over the normalised rows + Cosine::distance check for r < 1; the exact
    two-pass scan, device-driven, with the engine set to exact) and the call's asynchrony: it returns while work queued in
    front of it on its stream is still running"""
    import torch
    pts = (uniform((20000, 24), 6201) - np.float32(0.3)).astype(np.float32)
    qs = (uniform((100, 24), 6202) - np.float32(0.3)).astype(np.float32)
    tree = pn.BallTree.new(pts, pn.distance.Cosine())
    r = np.float32(0.2)
    want_off, want_idx = tree.query_radius_batch(qs, r)
    total = int(want_off[-1])
    assert total > 0
    off, idx, tot = _device_lists(tree, qs, r, total)
    assert tot == total and np.array_equal(off, want_off) and np.array_equal(idx[:total], want_idx)
    # asynchrony: a long kernel in front on the same stream; the call must return before it finishes
    big = torch.empty((8192, 8192), device="cuda:0")
    qd = torch.from_numpy(qs).to("cuda:0")
    torch.cuda.synchronize()
    ev = torch.cuda.Event()
    for _ in range(20):
        big = big @ big.clamp(-1e-3, 1e-3)
    ev.record()
    offs, idx2, tot2 = tree.query_radius_device(qd, r, total)
    returned_early = not ev.query()
    torch.cuda.synchronize()
    assert returned_early, "the device entry point waited for the stream"
    assert int(tot2.item()) == total and np.array_equal(idx2.cpu().numpy().astype(np.uint64)[:total], want_idx)
