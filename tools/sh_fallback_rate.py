"""Unproven-query rate of the bf16 tier over many DIFFERENT query batches (shared-threshold rank experiments).
usage: sh_fallback_rate.py <batches> <rank (PN_OPT_SHARED_THRESHOLDS value)> [k] [n_rows] [seed model 0/1]
(round 4: with the default options an index of this kind takes its starting thresholds from the seed model -- the rate
printed is then the model's; PN_EXP_MODEL_RANK moves where it aims)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import petal_neighbors_amd as pn
from petal_neighbors_amd import _lib
L = _lib.lib()
nb, rank = int(sys.argv[1]), int(sys.argv[2])
k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
n, dim, nq = (int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000), 128, 10_000
pts = torch.empty((n, dim), dtype=torch.float32, device='cuda:0'); qs = torch.empty((nq, dim), dtype=torch.float32, device='cuda:0')
L.pn_fill_uniform_device_f32(pts.data_ptr(), n * dim, 0x5EED0001, 0, 0, None)
t = pn.BallTree.from_device(pts)
t.set_option(_lib.PN_OPT_SHARED_THRESHOLDS, rank)
if len(sys.argv) > 5:
    t.set_option(_lib.PN_OPT_SEED_MODEL, int(sys.argv[5]))
tot = 0
for b in range(nb):
    L.pn_fill_uniform_device_f32(qs.data_ptr(), nq * dim, 0xABC0000 + b, 0, 0, None)
    t.query_device(qs, k)
torch.cuda.synchronize()
st = t.stats()
print("n %d seed model %s; rank %d k %d: %d queries, %d unproven (%.2e), candidates/query %.1f, evaluations/query %.1f" %
      (n, "accepted" if t.seed_model else "not accepted", rank, k, st["queries"], st["fallback_queries"], st["fallback_queries"] / max(st["queries"], 1),
       st["candidates"] / max(st["queries"], 1), st["evaluations"] / max(st["queries"], 1)))
