import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import petal_neighbors_amd as pn, oracle
from conftest import uniform
n, dim, r = 5000, 16, 1.2
pts = uniform((n, dim), 31 + n, np.float32)
qs = np.concatenate([pts[:20], uniform((13, dim), 32 + n, np.float32)])
for eng in ("bf16", "mfma", "exact"):
    t = pn.BallTree.euclidean(pts); t.set_engine(eng)
    off, idx = t.query_radius_batch(qs, r)
    want = [oracle.brute_radius(pts, qs[a], np.float32(r)) for a in range(len(qs))]
    bad = [a for a in range(len(qs)) if not np.array_equal(idx[int(off[a]):int(off[a+1])], want[a])]
    print(eng, "stats", t.stats(), "bad queries", bad[:10], "len got/want q0", int(off[1]-off[0]), len(want[0]))
for rr in (0.5, 0.8):
    t = pn.BallTree.euclidean(pts); t.set_engine("bf16")
    off, idx = t.query_radius_batch(qs, rr)
    want = [oracle.brute_radius(pts, qs[a], np.float32(rr)) for a in range(len(qs))]
    bad = [a for a in range(len(qs)) if not np.array_equal(idx[int(off[a]):int(off[a+1])], want[a])]
    print("r", rr, "bad", bad[:10], [ (int(off[a+1]-off[a]), len(want[a])) for a in range(5)], t.stats()["fallback_queries"])
