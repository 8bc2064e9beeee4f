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
      "per slow call: %.0f ns = count %.0f + append %.0f + compaction check %.0f;" % tuple(10 * out[i] / max(out[0], 1) for i in (3, 5, 6, 7)),
      "per compaction %.2f us; per wave: slow path %.2f ms of which compactions %.2f ms" % (10e-3 * out[7] / max(out[2], 1), 10e-6 * out[3] / 2048, 10e-6 * out[7] / 2048),
      "")
