import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import oracle, petal_neighbors_amd as pn
from petal_neighbors_amd import _lib
L = _lib.lib()
n, dim = 1_000_000, 128
pts = torch.empty((n, dim), dtype=torch.float32, device="cuda:0")
L.pn_fill_uniform_device_f32(pts.data_ptr(), pts.numel(), 0x5EED0001, 0, 0, None); torch.cuda.synchronize()
big = pn.BallTree.from_device(pts)
q = oracle.fill_uniform(64 * dim, 0x5EED0002).reshape(64, dim)
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for i in range(4): big.query_batch(q[:nq], 10)
