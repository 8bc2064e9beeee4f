"""Event counts of the bf16 filter (needs a PN_DIAG_FLAGS=-DPN_DIAG_BF_COUNT build).  usage: count_events_bf.py [slots] [k]"""
import sys, ctypes as C, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
waves = 480 * 4
ns = lambda i: 10.0 * out[i]   # s_memtime ticks of 10 ns
print("main launch only (the scout-only launch touches no buffers); per wave, of %d waves:" % waves)
print("  run time            %8.3f ms" % (ns(7) / waves * 1e-6))
print("  rare path (bf_slow) %8.3f ms in %.0f entries (%.0f ns each), %.1f appends per (segment, query) buffer" %
      (ns(3) / waves * 1e-6, out[0] / waves, ns(3) / max(out[0], 1), out[1] / (480 * 256)))
print("  ... of which mid-run compactions %8.3f ms in %.1f compactions (%.2f us each)" %
      (ns(4) / waves * 1e-6, out[2] / waves, ns(4) / max(out[2], 1) * 1e-3))
print("  end of run (final compactions + publish) %8.3f ms" % (ns(5) / waves * 1e-6))
print("  tile barrier incl. DMA wait %8.3f ms" % (ns(6) / waves * 1e-6))
