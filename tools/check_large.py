"""Ad-hoc parity probe at configs[2] / configs[3] scale: GPU answers for a sample of queries against the oracle's brute
force on the same device-generated corpus (test infrastructure: uses oracle/).  usage: check_large.py n dim nq k [sample]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import oracle
import petal_neighbors_amd as pn
from petal_neighbors_amd import _lib
L = _lib.lib()
n, dim, nq, k = (int(x) for x in sys.argv[1:5])
ns = int(sys.argv[5]) if len(sys.argv) > 5 else 12
pts = torch.empty((n, dim), dtype=torch.float32, device='cuda:0'); qs = torch.empty((nq, dim), dtype=torch.float32, device='cuda:0')
L.pn_fill_uniform_device_f32(pts.data_ptr(), n * dim, 0x5EED0001, 0, 0, None); L.pn_fill_uniform_device_f32(qs.data_ptr(), nq * dim, 0x5EED0002, 0, 0, None)
torch.cuda.synchronize()
t = pn.BallTree.from_device(pts)
idx = torch.empty((nq, k), dtype=torch.int64, device='cuda:0'); dist = torch.empty((nq, k), dtype=torch.float32, device='cuda:0')
t.query_device(qs, k, idx, dist); torch.cuda.synchronize()
t0 = time.perf_counter(); t.query_device(qs, k, idx, dist); torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = t.stats()
sel = np.linspace(0, nq - 1, ns).astype(np.int64)
ph = pts.cpu().numpy(); qh = qs[torch.as_tensor(sel, device='cuda:0')].cpu().numpy()
oi, od = oracle.brute_knn(ph, qh, k)
gi = idx[torch.as_tensor(sel, device='cuda:0')].cpu().numpy(); gd = dist[torch.as_tensor(sel, device='cuda:0')].cpu().numpy()
ok = gd.tobytes() == od.tobytes() and np.array_equal(gi.astype(np.uint64), oi.astype(np.uint64))
print(f"n={n} D={dim} nq={nq} k={k}: {dt*1e3:.1f} ms per batch, fallbacks {st['fallback_queries']}, sample of {ns} queries vs oracle: {'IDENTICAL' if ok else 'MISMATCH'}")
sys.exit(0 if ok else 1)
