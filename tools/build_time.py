"""Index build time from rows resident in HBM (pn_index_create_device_f32), incl. the seed model's moments + calibration
(round 4).  usage: build_time.py [n] [dim]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import petal_neighbors_amd as pn
from petal_neighbors_amd import _lib
L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
pts = torch.empty((n, dim), dtype=torch.float32, device='cuda:0')
L.pn_fill_uniform_device_f32(pts.data_ptr(), n * dim, 0x5EED0001, 0, 0, None)
torch.cuda.synchronize()
ts = []
for i in range(4):
    t0 = time.perf_counter()
    t = pn.BallTree.from_device(pts)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
    sm = t.seed_model
    t.close()
print(f"{n} x {dim} f32: build {min(ts[1:]):.2f} ms (first {ts[0]:.1f} ms), seed model {'accepted' if sm else 'not accepted'}")
