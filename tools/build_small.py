"""Index build latency for a small corpus (benches/ball_tree.rs 'build' shape).  usage: build_small.py [n] [dim] [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import petal_neighbors_amd as pn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 10
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
pts = np.random.default_rng(1).random((n, dim))
for _ in range(5):
    t = pn.BallTree.euclidean(pts); del t
t0 = time.perf_counter()
for _ in range(reps):
    t = pn.BallTree.euclidean(pts); del t
print("build+destroy %dx%d f64: %.1f us" % (n, dim, (time.perf_counter() - t0) / reps * 1e6))
keep = []
t0 = time.perf_counter()
for _ in range(reps):
    keep.append(pn.BallTree.euclidean(pts))
print("build only: %.1f us" % ((time.perf_counter() - t0) / reps * 1e6))
