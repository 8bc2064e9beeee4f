"""The *_device entry points are asynchronous (include/petal_mi355x.h, "Asynchronous device API"): a call enqueues its
work on the caller's stream and returns -- no hipStreamSynchronize, no read-back of the unproven-query count (round 1
had one).  Observable from outside: put something slow on the stream first; the call must come back while that is
still running, and the answers must still be the oracle's once the stream has drained."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _busy(torch, stream, ms_target=150.0):
    """enqueue roughly ms_target of matrix products on `stream`; returns an event recorded behind them"""
    a = torch.rand((4096, 4096), device="cuda")
    b = torch.rand((4096, 4096), device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        c = a @ b
    torch.cuda.synchronize()
    one = max((time.perf_counter() - t0) * 1e3, 0.05)
    reps = int(min(max(ms_target / one, 8), 4000))
    with torch.cuda.stream(stream):
        for _ in range(reps):
            c = a @ b
        ev = torch.cuda.Event()
        ev.record(stream)
    return ev, c


@pytest.mark.parametrize("engine", ["auto", "exact"])
def test_query_device_returns_before_its_stream_has_drained(pn, oracle_mod, engine):
    import torch
    rng = np.random.default_rng(5)
    pts = rng.random((60000, 64), dtype=np.float32)
    qs = rng.random((700, 64), dtype=np.float32)
    qs[3] = pts[17]          # an exact hit
    qs[5, 0] = np.nan        # a query only the second tier can answer: the call must not wait to learn that
    tree = pn.BallTree.euclidean(pts)
    tree.set_engine(engine)
    dq = torch.from_numpy(qs).cuda()
    tree.query_device(dq, 10)            # warm-up: workspaces, plans
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    ev, keep = _busy(torch, stream)
    assert not ev.query(), "the stream drained before the test could look (box too fast for the filler)"
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        idx, dist = tree.query_device(dq, 10)
    dt = (time.perf_counter() - t0) * 1e3
    still_busy = not ev.query()
    assert still_busy, f"pn_query_device_f32 came back only after the work queued in front of it had finished ({dt:.1f} ms)"
    assert dt < 50.0, f"enqueueing took {dt:.1f} ms"
    stream.synchronize()
    oidx, odist = oracle_mod.brute_knn(pts, qs, 10)
    assert dist.cpu().numpy().tobytes() == odist.tobytes()
    assert np.array_equal(idx.cpu().numpy().astype(np.uint64), oidx)
    del keep


def test_sharded_query_device_is_asynchronous_too(pn, oracle_mod):
    import torch
    rng = np.random.default_rng(6)
    pts = rng.random((50000, 32), dtype=np.float32)
    qs = rng.random((513, 32), dtype=np.float32)
    sh = pn.ShardedIndex.from_host(pts, devices=[0, 0, 0])     # three virtual shards: local merge, no exchange
    dq = torch.from_numpy(qs).cuda()
    sh.query_device(dq, 7)
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    ev, keep = _busy(torch, stream)
    assert not ev.query()
    with torch.cuda.stream(stream):
        idx, dist = sh.query_device(dq, 7)
    assert not ev.query(), "pn_sharded_query_device_f32 waited for its stream"
    stream.synchronize()
    oidx, odist = oracle_mod.brute_knn(pts, qs, 7)
    assert dist.cpu().numpy().tobytes() == odist.tobytes()
    assert np.array_equal(idx.cpu().numpy().astype(np.uint64), oidx)
    del keep
