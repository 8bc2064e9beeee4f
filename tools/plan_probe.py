#!/usr/bin/env python3
"""What the bf16 tier does with a (corpus, batch, k): per call the plan (PN_DEBUG_PLAN=1 on stderr) and the statistics.
usage (GPU box): PN_DEBUG_PLAN=1 python tools/plan_probe.py n dim nq k [calls]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import oracle
import petal_neighbors_amd as pn
from petal_neighbors_amd import _lib
n, dim, nq, k = (int(x) for x in sys.argv[1:5])
calls = int(sys.argv[5]) if len(sys.argv) > 5 else 3
L = _lib.lib()
pts = torch.empty((n, dim), dtype=torch.float32, device="cuda:0")
qs = torch.empty((nq, dim), dtype=torch.float32, device="cuda:0")
assert L.pn_fill_uniform_device_f32(pts.data_ptr(), n * dim, 0x5EED0001, 0, 0, None) == 0
assert L.pn_fill_uniform_device_f32(qs.data_ptr(), nq * dim, 0x5EED0002, 0, 0, None) == 0
tree = pn.BallTree.from_device(pts)
tree.set_option(_lib.PN_OPT_PROFILE, 2)
for c in range(calls):
    tree.stats(reset=True)
    i, d = tree.query_device(qs, k)
    torch.cuda.synchronize()
    st = tree.stats()
    print(f"call {c}: fallback {st['fallback_queries']} of {st['queries']}, candidates/q {st['candidates'] / max(st['queries'], 1):.1f}, "
          f"evaluations/q {st['evaluations'] / max(st['queries'], 1):.1f}, hot_ms {st['hot_ms']:.3f}, call ms {st['last_call_ms']:.3f}", flush=True)
