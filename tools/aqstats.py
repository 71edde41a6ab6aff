import sys, torch
sys.path.insert(0, "/root/repo")
import rayuela_jl_amd as rq
from rayuela_jl_amd import device as rqd, _lib
m=int(sys.argv[1]); d=int(sys.argv[2]); n=1000000; nq=10000; K=1000
g=torch.Generator(device="cuda").manual_seed(1)
codes=rqd.synth_codes(n,m,seed=1234)
cb=torch.randn((m*256,d),generator=g,device="cuda")
q=torch.randn((nq,d),generator=g,device="cuda")
norms=torch.rand((n,),generator=g,device="cuda")*100
rq.set_tuning("SCAN_STATS",1)
import inspect
for _ in range(2):
    out=rqd.linscan_aq(codes,cb,q,K,dbnorms=norms)
torch.cuda.synchronize()
st=_lib.scan_stats(); tot=sum(st[k] for k in ("lut","sample","stream","final_cut","sort_write"))
print({k:round(100*st[k]/tot,1) for k in ("lut","sample","stream","cuts","final_cut","sort_write")}, st["n_cuts"], st["n_fallbacks"], st["n_items"])
