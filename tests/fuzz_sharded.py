"""Randomised parity sweep of the row-sharded handle on one GPU (virtual shards: devices = [0] * G, one RCCL rank, the
exchange forced): k-NN and radius against the oracle's brute force (Euclidean, f32 and f64) or against the single
Cosine index.  Test infrastructure (it uses the oracle): run by tests/test_gpu_fuzz.py, or by hand:
python tests/fuzz_sharded.py [n_cases] [seed]"""
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle  # noqa: E402
import petal_neighbors_amd as pn  # noqa: E402
from petal_neighbors_amd import _lib  # noqa: E402
from petal_neighbors_amd.distance import Cosine  # noqa: E402


def run_case(c, rng):
    n = int(rng.choice([7, 100, 1000, 5000, 20000]))
    dim = int(rng.choice([3, 16, 64, 128, 200]))
    nq = int(rng.choice([1, 9, 70, 300]))
    k = int(rng.choice([1, 5, 10, 40]))
    shards = int(rng.choice([1, 2, 3, 7, 8]))
    f64 = bool(rng.integers(0, 2))
    cosine = bool(rng.integers(0, 4) == 0)
    dt = np.float64 if f64 else np.float32
    pts = rng.random((n, dim)).astype(dt)
    if n > 50:
        pts[n // 2] = pts[1]          # a duplicate across shards: ties broken by the GLOBAL index
    qs = np.concatenate([pts[:1], rng.random((nq, dim)).astype(dt)])[:nq] if nq > 1 else pts[:1].copy()
    sh = pn.ShardedIndex.from_host(pts, [0] * shards, metric=Cosine() if cosine else None)
    sh.set_option(_lib.PN_OPT_EXCHANGE_ALWAYS, 1)
    if cosine:
        one = pn.BallTree.new(pts, Cosine())
        wi, wd = one.query_batch(qs, k)
    else:
        wi, wd = oracle.brute_knn(pts, qs, k)
    gi, gd = sh.query_batch(qs, k)
    ok = np.array_equal(gi.astype(np.uint64), np.asarray(wi).astype(np.uint64)) and gd.tobytes() == np.ascontiguousarray(wd).tobytes()
    r = float(np.median(np.asarray(wd)[:, min(k, n) - 1])) if min(k, n) else 0.0
    off, ids = sh.query_radius_batch(qs, r)
    if cosine:
        o1, i1 = one.query_radius_batch(qs, r)
        ok = ok and np.array_equal(off, o1) and np.array_equal(ids, i1)
    else:
        for a in range(len(qs)):
            ok = ok and np.array_equal(ids[int(off[a]):int(off[a + 1])], oracle.brute_radius(pts, qs[a], dt(r)))
    sh.close()
    print(f"case {c}: n={n} D={dim} nq={nq} k={k} shards={shards} {'f64' if f64 else 'f32'} {'cosine' if cosine else 'euclidean'}: "
          f"{'ok' if ok else 'MISMATCH'}", flush=True)
    return ok


def main(n_cases=40, seed=1):
    rng = np.random.default_rng(seed)
    bad = sum(0 if run_case(c, rng) else 1 for c in range(n_cases))
    print("mismatches:", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
