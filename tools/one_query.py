import sys, os, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import oracle, petal_neighbors_amd as pn
from petal_neighbors_amd import _lib
L = _lib.lib()
n, dim = 1_000_000, 128
pts = torch.empty((n, dim), dtype=torch.float32, device="cuda:0")
L.pn_fill_uniform_device_f32(pts.data_ptr(), pts.numel(), 0x5EED0001, 0, 0, None); torch.cuda.synchronize()
big = pn.BallTree.from_device(pts)
q = oracle.fill_uniform(64 * dim, 0x5EED0002).reshape(64, dim)
for nq in (1, 8, 32):
    big.stats(reset=True)
    for i in range(3): big.query_batch(q[:nq], 10)
    t0 = time.perf_counter()
    for i in range(20): r = big.query_batch(q[:nq], 10)
    dt = (time.perf_counter() - t0) / 20
    st = big.stats()
    print("nq", nq, "us/call %.1f" % (dt * 1e6), "fallback", st["fallback_queries"], "cand/q %.1f" % (st["candidates"] / max(st["queries"], 1)), "queries", st["queries"])
    big.set_option(_lib.PN_OPT_PROFILE, 1)
    big.stats(reset=True)
    for i in range(10): big.query_batch(q[:nq], 10)
    st = big.stats()
    print("   hot ms per call %.4f (launches %d)" % (st["hot_ms"] / 10, st["hot_launches"]))
    big.set_option(_lib.PN_OPT_PROFILE, 0)
pts_h = oracle.fill_uniform(n * dim, 0x5EED0001).reshape(n, dim)
wi, wd = oracle.brute_knn(pts_h, q[:8], 10)
gi, gd = big.query_batch(q[:8], 10)
print("parity nq=8:", np.array_equal(gi, wi) and gd.tobytes() == wd.tobytes())
