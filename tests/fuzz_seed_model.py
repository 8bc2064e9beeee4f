"""Randomised sweep of LARGE indexes (10^5 .. 6 10^5 rows: where the bf16 tier's seed model is fitted):
the bf16 tier with the model on against the exact engine of the same index on every query (bit for bit), and against the
oracle's brute force on a few.  Test infrastructure (it uses the oracle): run by tests/test_gpu_fuzz.py, or by hand:
python tests/fuzz_seed_model.py [n_cases] [seed]"""
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle  # noqa: E402
import petal_neighbors_amd as pn  # noqa: E402
from conftest import uniform  # noqa: E402

FAMILIES = ["uniform", "centered", "gaussian", "scaled", "clusters_loose", "clusters_tight", "dups", "sorted", "outliers",
            "lowrank"]


def make_points(kind, n, dim, c, rng):
    if kind == "uniform":
        return uniform((n, dim), 1000 + c)
    if kind == "centered":
        return uniform((n, dim), 1000 + c) - np.float32(0.5)
    if kind == "gaussian":
        return rng.standard_normal((n, dim), dtype=np.float32)
    if kind == "scaled":  # per-dimension scales and offsets
        sc = rng.uniform(0.05, 3.0, size=dim).astype(np.float32)
        return rng.standard_normal((n, dim), dtype=np.float32) * sc + rng.uniform(-2, 2, size=dim).astype(np.float32)
    if kind in ("clusters_loose", "clusters_tight"):
        nc = int(rng.choice([5, 40, 300]))
        cen = rng.standard_normal((nc, dim), dtype=np.float32) * np.float32(2.0)
        w = (rng.uniform(0.3, 1.0, size=nc) if kind == "clusters_loose" else rng.uniform(0.005, 0.3, size=nc)).astype(np.float32)
        lab = rng.integers(0, nc, n)
        return cen[lab] + rng.standard_normal((n, dim), dtype=np.float32) * w[lab, None]
    if kind == "dups":
        p = uniform((n, dim), 1000 + c)
        p[n // 2:] = p[: n - n // 2]
        return p
    if kind == "sorted":
        p = uniform((n, dim), 1000 + c)
        return np.ascontiguousarray(p[np.argsort(p[:, 0])])
    if kind == "outliers":  # a handful of rows a thousand times farther out: they own the fourth moments
        p = uniform((n, dim), 1000 + c)
        p[rng.integers(0, n, 20)] *= np.float32(1000.0)
        return p
    if kind == "lowrank":  # rows near a 4-dimensional subspace
        b = rng.standard_normal((4, dim), dtype=np.float32)
        return rng.standard_normal((n, 4), dtype=np.float32) @ b + np.float32(0.01) * rng.standard_normal((n, dim), dtype=np.float32)
    raise ValueError(kind)


def run_case(c, rng, verbose=True):
    from petal_neighbors_amd import _lib
    n = int(rng.choice([100_000, 130_000, 250_000, 400_000, 600_000]))
    dim = int(rng.choice([8, 16, 64, 96, 128, 128, 200, 768]))
    if dim > 128:
        n = min(n, 250_000)
    nq = int(rng.choice([256, 1000, 3000]))
    k = int(rng.choice([1, 5, 10, 10, 33, 100, 128]))
    kind = str(rng.choice(FAMILIES))
    f64 = bool(rng.integers(0, 4) == 0)
    cosine = bool(rng.integers(0, 4) == 0) and kind != "outliers"
    qkind = str(rng.choice(["like", "near_rows", "far", "one_spot"]))
    pts = np.ascontiguousarray(make_points(kind, n, dim, c, rng), dtype=np.float32)
    if qkind == "like":
        qs = make_points(kind, nq, dim, c + 50000, rng)
    elif qkind == "near_rows":
        qs = pts[rng.integers(0, n, nq)] + np.float32(0.01) * rng.standard_normal((nq, dim), dtype=np.float32)
    elif qkind == "far":
        qs = make_points(kind, nq, dim, c + 50000, rng) * np.float32(3.0) + np.float32(1.5)
    else:
        qs = np.repeat(pts[rng.integers(0, n, 1)], nq, axis=0) + np.float32(1e-3) * rng.standard_normal((nq, dim), dtype=np.float32)
    qs = np.ascontiguousarray(qs, dtype=np.float32)
    if f64:
        pts = pts.astype(np.float64) + uniform((n, dim), 31000 + c).astype(np.float64) * 2.0 ** -26
        qs = qs.astype(np.float64) + uniform((nq, dim), 32000 + c).astype(np.float64) * 2.0 ** -26
    t = pn.BallTree.new(pts, pn.distance.Cosine()) if cosine else pn.BallTree.euclidean(pts)
    if not t.bf16_eligible:
        t.close()
        return True
    t.set_engine("exact")
    ei, ed = t.query_batch(qs, k)
    t.set_engine("bf16")
    ok = True
    seen = []
    for rep in range(2):  # (the second call may run after a sticky widening / switch-off)
        t.stats(reset=True)
        gi, gd = t.query_batch(qs, k)
        st = t.stats(reset=True)
        seen.append(st["fallback_queries"])
        nan = np.isnan(ed)
        same = np.array_equal(np.isnan(gd), nan) and gd[~nan].tobytes() == ed[~nan].tobytes() and np.array_equal(gi[~nan], ei[~nan])
        ok = ok and same
    t.set_option(_lib.PN_OPT_SEED_MODEL, 0)
    oi, od = t.query_batch(qs, k)
    nan = np.isnan(ed)
    ok = ok and od[~nan].tobytes() == ed[~nan].tobytes() and np.array_equal(oi[~nan], ei[~nan])
    # query_radius through the tier (Euclidean; Cosine for r < 1) against the exact scan, at a stored distance's value
    rr = ed[0][np.isfinite(ed[0])]
    if len(rr) and rr[-1] > 0 and (not cosine or rr[-1] < 1):
        r = float(rr[-1])
        sub = np.ascontiguousarray(qs[:200])
        t.set_engine("exact")
        xo, xi = t.query_radius_batch(sub, r)
        t.set_engine("bf16")
        go, gi = t.query_radius_batch(sub, r)
        ok = ok and np.array_equal(go, xo) and np.array_equal(gi, xi)
    if not cosine:  # the oracle's brute force on a few queries
        sel = rng.choice(nq, 4, replace=False)
        bi, bd = oracle.brute_knn(pts, qs[sel], k)
        ok = ok and bd.tobytes() == ed[sel].tobytes() and np.array_equal(bi, ei[sel])
    if verbose or not ok:
        print(f"case {c}: n={n} dim={dim} nq={nq} k={k} {kind} q={qkind} f64={f64} cosine={cosine} model={t.seed_model} "
              f"unproven={seen} {'OK' if ok else 'MISMATCH'}", flush=True)
    t.close()
    return ok


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(n_cases):
        bad += 0 if run_case(1000 * seed + c, rng) else 1
    print(f"{n_cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)
