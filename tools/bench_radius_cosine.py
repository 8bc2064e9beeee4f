"""query_radius on a Cosine index (round 4): the bf16 tier against the exact scan, host and device API, 1M x 128 uniform
[-0.5, 0.5) (PN_RADIUS_N / PN_RADIUS_NQ: other sizes), r = the median nearest-neighbour Cosine distance x 1.0005.
usage: bench_radius_cosine.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import petal_neighbors_amd as pn
from petal_neighbors_amd import _lib
n, dim, nq = int(os.environ.get('PN_RADIUS_N', '1000000')), 128, int(os.environ.get('PN_RADIUS_NQ', '10000'))
rng = np.random.default_rng(1)
pts = (rng.random((n, dim), dtype=np.float32) - np.float32(0.5))
qs = (rng.random((nq, dim), dtype=np.float32) - np.float32(0.5))
t = pn.BallTree.new(pts, pn.distance.Cosine())
_, d1 = t.query_batch(qs[:2000], 1)
r = float(np.float32(np.median(d1[:, 0]) * 1.0005))
qd = torch.from_numpy(qs).to("cuda:0")
res = {}
for eng in ("bf16", "exact"):
    t.set_engine(eng)
    off, idx = t.query_radius_batch(qs, r)
    reps = 5 if eng == "bf16" else 2
    t0 = time.perf_counter()
    for _ in range(reps):
        off, idx = t.query_radius_batch(qs, r)
    dt = (time.perf_counter() - t0) / reps
    o, i, tot = t.query_radius_device(qd, r, len(idx) + 8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        o, i, tot = t.query_radius_device(qd, r, len(idx) + 8, out_offsets=o, out_idx=i, out_total=tot)
    torch.cuda.synchronize()
    dd = (time.perf_counter() - t0) / reps
    res[eng] = (off.copy(), idx.copy())
    print(f"Cosine query_radius {n} x {dim}, {nq} queries, r={r:.6f} ({len(idx)} results) engine={eng}: host API {dt*1e3:.2f} ms "
          f"({nq/dt/1e6:.3f} M q/s), device API {dd*1e3:.2f} ms ({nq/dd/1e6:.3f} M q/s), fallbacks {t.stats(reset=True)['fallback_queries']}")
print("lists identical:", np.array_equal(res["bf16"][0], res["exact"][0]) and np.array_equal(res["bf16"][1], res["exact"][1]))
