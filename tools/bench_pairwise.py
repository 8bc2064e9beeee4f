#!/usr/bin/env python3
"""distance::pairwise(x, &Euclidean) at scale (reference src/distance.rs:58-74) through pn_pairwise_device_*:
time per call with rows and the n x n result resident in HBM, priced against the two roofs the kernel can hit.

 * VALU: the reference fold is 3 separately rounded operations per coordinate and pair (sub, mul, add; no fma by
   contract), evaluated for the n(n-1)/2 pairs above the diagonal only -> 3*D*n(n-1)/2 operations.  Peak used:
   the guide's non-fma vector rate = half the fma FLOP/s figure
   (f32 157.3/2 TFLOP/s packed, f64 78.6/2).
 * HBM: the result is written twice (both triangles): n*n*sizeof(T) bytes + the rows once.

One JSON line per shape; `python tools/bench_pairwise.py > profiles/rNN_pairwise.json` on the GPU box.
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import petal_neighbors_amd as pn
from petal_neighbors_amd import distance as dist

PEAK_OPS = {torch.float32: 157.3e12 / 2, torch.float64: 78.6e12 / 2}
HBM = 8.0e12


def one(n, d, dtype, reps=5):
    g = torch.Generator(device="cuda").manual_seed(n * 131 + d)
    x = torch.rand((n, d), generator=g, device="cuda", dtype=dtype)
    out = torch.empty((n, n), device="cuda", dtype=dtype)
    dist.pairwise_device(x, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dist.pairwise_device(x, out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    # spot checks: symmetry, zero diagonal, 64 entries against the scalar metric
    assert torch.equal(out, out.T) and float(out.diagonal().abs().max()) == 0.0
    xs = x.cpu().numpy()
    o = out.cpu().numpy() if n <= 8192 else None
    eu = dist.Euclidean()
    rng = torch.Generator().manual_seed(7)
    ii = torch.randint(0, n, (64, 2), generator=rng).numpy()
    for i, j in ii:
        v = o[i, j] if o is not None else float(out[int(i), int(j)])
        assert v == eu.distance(xs[i], xs[j]) or i == j, (i, j)
    es = x.element_size()
    ops = 3.0 * d * n * (n - 1) / 2
    byts = n * n * es + n * d * es
    valu_ms = ops / PEAK_OPS[dtype] * 1e3
    hbm_ms = byts / HBM * 1e3
    bound = "valu" if valu_ms > hbm_ms else "hbm"
    return {"workload": f"pairwise {n} x {d} {'f32' if es == 4 else 'f64'}", "ms": round(ms, 3),
            "pairs_per_s": round(n * (n - 1) / 2 / ms * 1e3, 1), "bound": bound,
            "valu_roof_ms": round(valu_ms, 3), "hbm_roof_ms": round(hbm_ms, 3),
            "frac": round(max(valu_ms, hbm_ms) / ms, 4),
            "achieved_ops_T": round(ops / ms / 1e9, 2), "achieved_write_GBs": round(byts / ms / 1e6, 1)}


def main():
    for dtype in (torch.float32, torch.float64):
        for n, d in ((16384, 16), (16384, 128), (8192, 768), (32768, 128)):
            print(json.dumps(one(n, d, dtype)), flush=True)


if __name__ == "__main__":
    main()
