import sys, numpy as np
sys.path.insert(0,'/root/repo')
import rayuela_jl_amd.synth as synth
from oracle import oracle
n,d,m,h=200_000,128,8,256
nq=64
X=synth.sift_like(n,d,seed=synth.SEED_BASE,ncentres=65536)
Q=synth.sift_like(nq,d,seed=synth.SEED_BASE,ncentres=65536,row0=3_000_000_000)
S=synth.sift_like(20000,d,seed=synth.SEED_BASE,ncentres=65536,row0=3_100_000_000)
C=synth.codebooks(S,m,h,seed=synth.SEED_CODEBOOK,iters=5,sample=20000)
codes=oracle.encode_pq(X,synth.cat_codebooks(C),m,h)
cen=np.stack(C)
for quant in (1e-3,1e-2):
    K=int(n*quant)
    res={}
    alive_full=np.zeros(n,bool); 
    for g0 in range(0,nq,8):
        af=np.zeros(n,bool); ap={t:np.zeros(n,bool) for t in (4,5,6)}
        for q in range(g0,g0+8):
            lut=oracle.adc_lut(cen,Q[q]).reshape(m,256)
            dist=np.zeros(n,np.float32)
            for k in range(m): dist=dist+lut[k][codes[:,k]]
            tau=np.partition(dist,int(K*1.65))[int(K*1.65)]   # ~1.65K survivors like the kernel's tau after retune
            mins=lut.min(1); base=mins.sum()
            THR=95.0; inv=THR/(tau-base)
            e=np.minimum(np.floor((lut-mins[:,None])*inv),27).astype(np.int32)
            sums=np.zeros(n,np.int32)
            part={}
            for k in range(m):
                sums=sums+e[k][codes[:,k]]
                if k+1 in (4,5,6): part[k+1]=sums.copy()
            af|= sums<=95
            for t in (4,5,6): ap[t]|= part[t]<=95
        res.setdefault('full',[]).append(af.mean())
        for t in (4,5,6): res.setdefault('first%d'%t,[]).append(ap[t].mean())
    print("quantile",quant,{k:round(float(np.mean(v)),4) for k,v in res.items()})
