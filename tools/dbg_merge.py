import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from conftest import uniform
import petal_neighbors_amd as pn
from petal_neighbors_amd.sharded import HipShardEngine, shard_bounds
import oracle
n,dim,nq,k,shards=20000,128,200,10,8
pts = uniform((n, dim), 77 + n, np.float32); pts[n-1]=pts[0]
qs = np.concatenate([pts[:3], uniform((nq - 3, dim), 78 + n, np.float32)])
qd = torch.from_numpy(qs).to("cuda:0")
k_part=10
idx_parts = torch.full((shards, nq, k_part), -1, dtype=torch.int64, device="cuda:0")
dst_parts = torch.full((shards, nq, k_part), float("nan"), dtype=torch.float32, device="cuda:0")
engs=[]
for r in range(shards):
    lo,hi=shard_bounds(n,shards,r)
    eng=HipShardEngine(0); eng.build(pts[lo:hi], lo); engs.append(eng)
    li,ld=eng.query(qd,k_part)
    torch.cuda.synchronize()
    wi,wd=oracle.brute_knn(pts[lo:hi],qs,k_part)
    ok = np.array_equal(li.cpu().numpy().astype(np.uint64), wi+np.uint64(lo)) and ld.cpu().numpy().tobytes()==wd.tobytes()
    print("shard",r,lo,hi,"local ok",ok, li[0,:3].tolist())
    idx_parts[r,:,:li.shape[1]]=li; dst_parts[r,:,:ld.shape[1]]=ld
mi,md=engs[-1].merge(idx_parts,dst_parts,10); torch.cuda.synchronize()
wi,wd=oracle.brute_knn(pts,qs,10)
print(mi[0].tolist()); print(wi[0].tolist()); print(md[0].tolist()); print(wd[0].tolist())
bad=np.where((mi.cpu().numpy().astype(np.uint64)!=wi).any(axis=1))[0]; print("bad queries",bad[:20], len(bad))
