"""Event counts of the bf16 filter (needs a PN_DIAG_FLAGS=-DPN_DIAG_BF_COUNT build).  usage: count_events_bf.py [slots] [k]"""
import sys, ctypes as C, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import petal_neighbors_amd as pn
from petal_neighbors_amd import _lib
L = _lib.lib()
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 0
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n, dim, nq = int(os.environ.get('PN_N', 1_000_000)), 128, 10_000
pts = torch.empty((n, dim), dtype=torch.float32, device='cuda:0'); qs = torch.empty((nq, dim), dtype=torch.float32, device='cuda:0')
L.pn_fill_uniform_device_f32(pts.data_ptr(), n * dim, 0x5EED0001, 0, 0, None); L.pn_fill_uniform_device_f32(qs.data_ptr(), nq * dim, 0x5EED0002, 0, 0, None)
torch.cuda.synchronize()
t = pn.BallTree.from_device(pts)
t.set_engine("bf16")
if slots: t.set_option(_lib.PN_OPT_FILTER_SLOTS, slots)
if os.environ.get('PN_SH') is not None: t.set_option(_lib.PN_OPT_SHARED_THRESHOLDS, int(os.environ['PN_SH']))
f = L.pn_debug_read_bf; f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_int]
out = (C.c_ulonglong * 16)()
t.query_device(qs, k); torch.cuda.synchronize(); f(out, 1)
t.query_device(qs, k); torch.cuda.synchronize(); f(out, 1)
waves = 480 * 4
cyc = lambda i: float(out[i])   # s_memtime ticks = shader cycles (MI355X_MICROARCH.md, constants table)
run = cyc(7) / waves
print("main launch only (the scout-only launch touches no buffers); means per wave, of %d waves, in shader cycles:" % waves)
print("  run                  %10.0f cycles" % run)
print("  rare path (bf_slow)  %10.0f cycles = %4.1f %% of the run, in %.0f entries (%.0f cycles each); %.1f appends per (segment, query) buffer" %
      (cyc(3) / waves, 100.0 * cyc(3) / waves / run, out[0] / waves, cyc(3) / max(out[0], 1), out[1] / (480 * 256)))
print("  ... of which mid-run compactions %10.0f cycles in %.1f compactions (%.0f cycles each)" %
      (cyc(4) / waves, out[2] / waves, cyc(4) / max(out[2], 1)))
print("  end of run (final compactions + publish) %10.0f cycles" % (cyc(5) / waves))
print("  tile barrier incl. DMA wait %10.0f cycles = %4.1f %% of the run" % (cyc(6) / waves, 100.0 * cyc(6) / waves / run))
print("  tile top to chain 1 (fragment + norm reads issued, LDS-DMA issue, check, rare path) %10.0f cycles = %4.1f %% of the run" % (cyc(8) / waves, 100.0 * cyc(8) / waves / run))
print("  chain 1 (16 MFMAs)  %10.0f cycles = %4.1f %% of the run (%.0f cycles per chain)" % (cyc(9) / waves, 100.0 * cyc(9) / waves / run, cyc(9) / waves / max(n / 128 / 12 * 2, 1)))
print("  chain 2 (16 MFMAs)  %10.0f cycles = %4.1f %% of the run (%.0f cycles per chain)" % (cyc(10) / waves, 100.0 * cyc(10) / waves / run, cyc(10) / waves / max(n / 128 / 12 * 2, 1)))
print("  in-kernel clock (s_memtime / s_memrealtime x 100 MHz) %.3f GHz; run = %.3f ms" % (cyc(7) / max(cyc(11), 1) * 0.1, cyc(11) / waves * 1e-5))

if out[12]:
    print("  refreshers: %d query visits, %d updates, %d passes over all refresher waves (%.1f per wave), %d rejected (a word changed)" %
          (out[12], out[13], out[14], out[14] / 128.0, out[15]))
print("  fallback queries %d, candidates per query %.1f" % (t.stats()["fallback_queries"], t.stats()["candidates"] / max(t.stats()["queries"], 1)))
