"""query_radius throughput (host API: queries up, CSR down), C2-shaped corpus (PN_RADIUS_DIM=768: wide rows; PN_RADIUS_N / PN_RADIUS_NQ: other sizes).
usage: bench_radius.py [r ...]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import petal_neighbors_amd as pn
from petal_neighbors_amd import _lib
L = _lib.lib()
n, dim, nq = int(os.environ.get('PN_RADIUS_N', '1000000')), int(os.environ.get('PN_RADIUS_DIM', '128')), int(os.environ.get('PN_RADIUS_NQ', '10000'))
pts = torch.empty((n, dim), dtype=torch.float32, device='cuda:0'); qs = torch.empty((nq, dim), dtype=torch.float32, device='cuda:0')
L.pn_fill_uniform_device_f32(pts.data_ptr(), n * dim, 0x5EED0001, 0, 0, None); L.pn_fill_uniform_device_f32(qs.data_ptr(), nq * dim, 0x5EED0002, 0, 0, None)
torch.cuda.synchronize()
t = pn.BallTree.from_device(pts)
qh = qs.cpu().numpy()
for r in [float(x) for x in sys.argv[1:]] or [0.5, 3.3, 3.5]:
    for eng in ("bf16", "mfma" if dim <= 128 else "exact"):
        t.set_engine(eng)
        off, idx = t.query_radius_batch(qh, r)
        t0 = time.perf_counter()
        for _ in range(5):
            off, idx = t.query_radius_batch(qh, r)
        dt = (time.perf_counter() - t0) / 5
        print(f"r={r} engine={eng}: {dt*1e3:.2f} ms per {nq} queries = {nq/dt/1e6:.2f} M q/s, {len(idx)} results, stats {t.stats(reset=True)['fallback_queries']} fallbacks")
