"""Where a re-rank wave's time goes (needs a PN_DIAG_FLAGS=-DPN_DIAG_RR_STAMP build).  usage: rerank_phases.py [k]"""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import petal_neighbors_amd as pn
from petal_neighbors_amd import _lib
L = _lib.lib()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n, dim, nq = int(os.environ.get("PN_N", 1_000_000)), 128, 10_000
pts = torch.empty((n, dim), dtype=torch.float32, device='cuda:0'); qs = torch.empty((nq, dim), dtype=torch.float32, device='cuda:0')
L.pn_fill_uniform_device_f32(pts.data_ptr(), n * dim, 0x5EED0001, 0, 0, None); L.pn_fill_uniform_device_f32(qs.data_ptr(), nq * dim, 0x5EED0002, 0, 0, None)
torch.cuda.synchronize()
t = pn.BallTree.from_device(pts)
f = L.pn_debug_read_rr; f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_int]
out = (C.c_ulonglong * 16)()
t.query_device(qs, k); torch.cuda.synchronize()
t.query_device(qs, k); torch.cuda.synchronize(); f(out, nq)
names = ["counts+thresholds", "gather candidates", "select k-th bound", "evaluate round 1", "select k-th distance",
         "evaluate round 2", "cut to k", "rank + write"]
tot = sum(out[i] for i in range(8))
for i, nm in enumerate(names):
    print("  %-22s %9.0f cycles per wave  %5.1f %%" % (nm, out[i] / nq, 100.0 * out[i] / max(tot, 1)))
print("  total %.0f cycles per wave" % (tot / nq))
