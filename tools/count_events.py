import sys, ctypes as C, numpy as np, torch
sys.path.insert(0,'/root/repo')
import petal_neighbors_amd as pn
from petal_neighbors_amd import _lib
L=_lib.lib()
n,dim,nq,k=1_000_000,128,10_000,10
pts=torch.empty((n,dim),dtype=torch.float32,device='cuda:0'); qs=torch.empty((nq,dim),dtype=torch.float32,device='cuda:0')
L.pn_fill_uniform_device_f32(pts.data_ptr(), n*dim, 0x5EED0001, 0, 0, None); L.pn_fill_uniform_device_f32(qs.data_ptr(), nq*dim, 0x5EED0002, 0, 0, None)
torch.cuda.synchronize()
t=pn.BallTree.from_device(pts)
f=L.pn_debug_read; f.restype=C.c_int; f.argtypes=[C.c_void_p, C.c_int]
out=(C.c_ulonglong*8)()
t.query_device(qs,k); torch.cuda.synchronize(); f(out,1)
t.query_device(qs,k); torch.cuda.synchronize(); f(out,1)
sb,cp,ap=out[0],out[1],out[2]
waves=256*4; tiles=1_250_000*4  # wave-tiles
nw=1024
print("per-wave cycles: total %.3gM  slow %.3gM  barrier %.3gM  stage-store %.3gM"%(out[6]/nw/1e6,out[3]/nw/1e6,out[4]/nw/1e6,out[5]/nw/1e6))
print("slow blocks",sb,"per wave-tile %.3f"%(sb/tiles),"compactions",cp,"per query-seg %.1f"%(cp/(10240*3.2)),"appends",ap,"per query %.1f"%(ap/10240))
