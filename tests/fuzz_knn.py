"""Randomised parity sweep on the GPU: auto engine (all tiers) vs the oracle's brute force, bit-exact.
Test infrastructure (it uses the oracle): run by tests/test_gpu_fuzz.py, or by hand: python tests/fuzz_knn.py [n_cases] [seed]"""
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle  # noqa: E402
import petal_neighbors_amd as pn  # noqa: E402
from conftest import uniform  # noqa: E402


def run_case(c, rng):
    n = int(rng.choice([64, 1000, 3000, 4096, 5000, 9000, 20000, 33000, 60000]))  # (<= 4096 with few queries: one-launch path)
    dim = int(rng.choice([3, 8, 15, 16, 31, 64, 96, 100, 128, 129, 200, 384, 768, 1024, 1536, 2048]))
    nq = int(rng.choice([1, 7, 64, 255, 256, 257, 600, 1300]))
    if dim > 128:  # wide rows (K-chunked bf16 kernel): keep the oracle's brute force to seconds
        n, nq = min(n, 20000), min(nq, 600)
    k = int(rng.choice([1, 2, 5, 10, 33, 100, 100, 300, 700]))  # (round 4: large k through more segments per query tile)
    if k > 100 and (dim > 128 or n < 3000):
        k = 100
    kind = rng.choice(["uniform", "centered", "clustered", "dups", "sorted"])
    f64 = bool(rng.integers(0, 3) == 0)  # a third of the cases: an f64 index (bf16 filter, f64 re-rank and second tier)
    pts = uniform((n, dim), 1000 + c)
    if kind == "centered":
        pts = pts - np.float32(0.5)
    elif kind == "clustered":
        cen = uniform((12, dim), 7 + c) * np.float32(4)
        pts = (cen[rng.integers(0, 12, n)] + np.float32(0.05) * (uniform((n, dim), 99 + c) - np.float32(0.5))).astype(np.float32)
    elif kind == "dups":
        pts[n // 2:] = pts[: n - n // 2]
    elif kind == "sorted":
        pts = pts[np.argsort(pts[:, 0])]
    qs = pts[rng.integers(0, n, nq)] + np.float32(0.01) * (uniform((nq, dim), 5000 + c) - np.float32(0.5)) if kind != "uniform" else uniform((nq, dim), 5000 + c)
    qs = np.ascontiguousarray(qs, dtype=np.float32)
    if f64:  # coordinates with more than 24 significant bits
        pts = pts.astype(np.float64) + uniform((n, dim), 31000 + c).astype(np.float64) * 2.0 ** -26
        if kind == "dups":
            pts[n // 2:] = pts[: n - n // 2]
        qs = qs.astype(np.float64) + uniform((nq, dim), 32000 + c).astype(np.float64) * 2.0 ** -26
    from petal_neighbors_amd import _lib
    t = pn.BallTree.euclidean(pts)
    waves = int(rng.choice([0, 4, 8]))  # (round 4: either main-pass kernel of the bf16 tier)
    t.set_option(_lib.PN_OPT_BF16_WAVES, waves)
    idx, dist = t.query_batch(qs, k)
    oi, od = oracle.brute_knn(pts, qs, k)
    ok = dist.tobytes() == od.tobytes() and np.array_equal(idx, oi)
    # round 4: the same corpus as a Cosine index -- the bf16 tier over the normalised rows against the exact scan of the same
    # index (bit for bit, NaN distances aside) and, where it is cheap, the oracle's pairwise(x, &Cosine)
    if c % 3 == 0 and n <= 33000:
        try:
            tc = pn.BallTree.new(pts, pn.distance.Cosine())
            kc = min(k, 100)
            ci, cd = tc.query_batch(qs, kc)
            tc.set_engine("exact")
            ei, ed = tc.query_batch(qs, kc)
            nan = np.isnan(ed)
            ok = ok and np.array_equal(np.isnan(cd), nan) and cd[~nan].tobytes() == ed[~nan].tobytes() and np.array_equal(ci[~nan], ei[~nan])
            # query_radius on the Cosine index (round 4: through the tier for r < 1): a radius exactly at and just above a
            # stored distance, tier against exact scan, host and device entry
            fin_d = ed[0][np.isfinite(ed[0])]
            if len(fin_d):
                ftc = np.float64 if f64 else np.float32
                for rc in (ftc(fin_d[-1]), ftc(fin_d[len(fin_d) // 2]) * ftc(1.000001)):
                    if not (0 < rc < 1):
                        continue
                    tc.set_engine("exact")
                    xo, xi = tc.query_radius_batch(qs, float(rc))
                    tc.set_engine("auto")
                    go, gi = tc.query_radius_batch(qs, float(rc))
                    ok = ok and np.array_equal(go, xo) and np.array_equal(gi, xi)
                    import torch
                    qdc = torch.from_numpy(np.ascontiguousarray(qs)).to("cuda:0")
                    do, di, dt = tc.query_radius_device(qdc, float(rc), int(xo[-1]) + 3)
                    torch.cuda.synchronize()
                    ok = ok and int(dt.item()) == int(xo[-1]) and np.array_equal(do.cpu().numpy().astype(np.uint64), xo) and \
                        np.array_equal(di.cpu().numpy().astype(np.uint64)[: int(xo[-1])], xi)
            if n <= 5000 and nq <= 64:
                dm = oracle.pairwise_cosine(np.vstack([qs, pts]))[:nq, nq:]
                for a in range(nq):
                    nn_ = np.isnan(dm[a])
                    order = np.lexsort((np.arange(n), np.where(nn_, np.inf, dm[a]), nn_))[:kc]
                    fin = ~np.isnan(dm[a][order])
                    ok = ok and np.array_equal(ci[a][fin], order[fin].astype(np.uint64)) and cd[a][fin].tobytes() == dm[a][order][fin].tobytes()
        except pn.PetalError:
            ok = False
    # radius queries through the same tiers: a radius just above / exactly at a stored distance (strict '<')
    nr = min(nq, 8)
    ft = np.float64 if f64 else np.float32
    for r in (ft(od[0, int(rng.integers(0, od.shape[1]))]) * ft(1.000001), ft(od[0, -1])):
        if not np.isfinite(r) or r <= 0:
            continue
        off, ids = t.query_radius_batch(qs[:nr], float(r))
        for a in range(nr):
            ok = ok and np.array_equal(ids[int(off[a]):int(off[a + 1])], oracle.brute_radius(pts, qs[a], r))
        # round 4: the device entry point (counts, scan and fill in HBM), with room for everything and with a short buffer
        import torch
        qd = torch.from_numpy(np.ascontiguousarray(qs[:nr])).to("cuda:0")
        total = int(off[-1])
        for cap in (total + 5, max(total // 2, 1)):
            do, di, dt = t.query_radius_device(qd, float(r), cap)
            torch.cuda.synchronize()
            m = min(cap, total)
            ok = ok and int(dt.item()) == total and np.array_equal(do.cpu().numpy().astype(np.uint64), off) and \
                np.array_equal(di.cpu().numpy().astype(np.uint64)[:m], ids[:m])
    st = t.stats()
    print(f"case {c}: n={n} D={dim} nq={nq} k={k} {kind}{' f64' if f64 else ''}: {'ok' if ok else 'MISMATCH'} fallback {st['fallback_queries']}/{st['queries']} cand/q {st['candidates']/max(st['queries'],1):.0f}", flush=True)
    return ok


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = sum(0 if run_case(c, rng) else 1 for c in range(cases))
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)
