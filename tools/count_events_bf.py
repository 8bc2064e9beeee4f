"""Event counts of the bf16 filter (needs a PN_DIAG_FLAGS=-DPN_DIAG_BF_COUNT build).  usage: count_events_bf.py [slots] [k]"""
import sys, ctypes as C, torch
sys.path.insert(0, '/root/repo')
import petal_neighbors_amd as pn
from petal_neighbors_amd import _lib
L = _lib.lib()
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 0
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n, dim, nq = 1_000_000, 128, 10_000
pts = torch.empty((n, dim), dtype=torch.float32, device='cuda:0'); qs = torch.empty((nq, dim), dtype=torch.float32, device='cuda:0')
L.pn_fill_uniform_device_f32(pts.data_ptr(), n * dim, 0x5EED0001, 0, 0, None); L.pn_fill_uniform_device_f32(qs.data_ptr(), nq * dim, 0x5EED0002, 0, 0, None)
torch.cuda.synchronize()
t = pn.BallTree.from_device(pts)
t.set_engine("bf16")
if slots: t.set_option(_lib.PN_OPT_FILTER_SLOTS, slots)
f = L.pn_debug_read_bf; f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_int]
out = (C.c_ulonglong * 8)()
t.query_device(qs, k); torch.cuda.synchronize(); f(out, 1)
t.query_device(qs, k); torch.cuda.synchronize(); f(out, 1)
units = 40 * 4 * 2 * 31250
print("slow calls", out[0], "(%.3f of wave-block-qb units)" % (out[0] / units), "appends", out[1], "per query %.1f" % (out[1] / 10240),
      "lanes with survivors per call %.2f" % (out[4] / max(out[0], 1)), "compactions", out[2],
      "ticks per slow call %.0f" % (out[3] / max(out[0], 1)), "= scan %.0f + append %.0f + compaction %.0f" %
      (out[5] / max(out[0], 1), out[6] / max(out[0], 1), out[7] / max(out[0], 1)),
      "slow ticks per wave %.3gM" % (out[3] / 2048 / 1e6))
