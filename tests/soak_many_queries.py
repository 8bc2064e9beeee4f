"""Soak (by hand: python tests/soak_many_queries.py on a GPU box): batches of 30-70 thousand queries -- grids of workgroups in rounds, many
query tiles -- against corpora of 120-300 thousand rows, the bf16 tier with the seed model against the exact engine on every
query, bit for bit.  Test infrastructure."""
import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import petal_neighbors_amd as pn
from conftest import uniform
bad = 0
for (n, dim, nq, k, f64, cosine) in [(300_000, 64, 60_000, 10, False, False), (200_000, 128, 50_000, 100, False, False),
                                     (150_000, 96, 70_000, 10, True, False), (250_000, 32, 45_000, 33, False, True),
                                     (120_000, 200, 30_000, 10, False, False)]:
    dt = np.float64 if f64 else np.float32
    pts = uniform((n, dim), 9100 + dim, dt) - dt(0.25)
    qs = uniform((nq, dim), 9200 + dim, dt) - dt(0.25)
    t = pn.BallTree.new(pts, pn.distance.Cosine()) if cosine else pn.BallTree.euclidean(pts)
    t.set_engine("exact"); ei, ed = t.query_batch(qs, k)
    t.set_engine("bf16"); t.stats(reset=True); gi, gd = t.query_batch(qs, k); st = t.stats(reset=True)
    ok = gd.tobytes() == ed.tobytes() and np.array_equal(gi, ei)
    bad += 0 if ok else 1
    print(f"n={n} dim={dim} nq={nq} k={k} f64={f64} cosine={cosine} model={t.seed_model} unproven={st['fallback_queries']} cand/q={st['candidates']/st['queries']:.1f} {'OK' if ok else 'MISMATCH'}", flush=True)
    t.close()
print("mismatches", bad)
