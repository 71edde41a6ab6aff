#!/usr/bin/env python3
"""Round 6 (EXPERIMENTS.md 9.7), CPU only: LDS-pass model of the greedy balance of the row order.  The bench codes (SIFT1M shape,
encoded by the oracle) are sorted by a key of `bits`, then the rows of every bucket are dealt to the lane groups the bucket holds
(capacities of boundary groups respected, arrival order, cost = sum over the free tables of the marginal (column load)^2, equal
bytes free) and the passes per row and lane group are compared with the plain sort.  m = 16: uniform random codes.
usage: python tools/sim_greedy_order.py [codes.npy]   (without a file the 1e6 x 8 bench codes are generated first: ~1 min)"""
import os
import sys

sys.path.insert(0, os.getcwd())


def bench_codes():
    import rayuela_jl_amd.synth as synth
    from oracle import oracle
    n, d, m, h = 1_000_000, 128, 8, 256
    S = synth.sift_like(20000, d, seed=synth.SEED_BASE, ncentres=65536, row0=3_100_000_000)
    C = synth.codebooks(S, m, h, seed=synth.SEED_CODEBOOK, iters=5, sample=20000)
    Cc = synth.cat_codebooks(C)
    out = np.empty((n, m), np.uint8)
    for a in range(0, n, 250000):
        out[a:a + 250000] = oracle.encode_pq(synth.sift_like(250000, d, seed=synth.SEED_BASE, ncentres=65536, row0=a), Cc, m, h)
    return out


import time

import numpy as np
rng=np.random.default_rng(0)
def group_passes(g):
    G,_,m=g.shape; out=np.zeros((G,m))
    for k in range(m):
        b=g[:,:,k].astype(np.int64)
        pres=np.zeros((G,256),bool); pres[np.arange(G)[:,None],b]=True
        out[:,k]=pres.reshape(G,8,32).sum(1).max(1)
    return out
def run(codes,bits,tables,nb_sample,label):
    n,m=codes.shape
    key=np.zeros(n,np.uint64)
    for k,nb in enumerate(bits):
        if nb: key=(key<<np.uint64(nb))|(codes[:,k].astype(np.uint64)>>np.uint64(8-nb))
    o=np.argsort(key,kind='stable'); ks=key[o]
    bnd=np.flatnonzero(np.r_[True,ks[1:]!=ks[:-1],True])
    # pick a contiguous run of buckets so that boundary groups are realistic
    nbk=len(bnd)-1
    start=rng.integers(0,max(1,nbk-nb_sample)); sel=range(start,min(nbk,start+nb_sample))
    s_lo=bnd[start]; s_hi=bnd[min(nbk,start+nb_sample)]
    newpos=np.empty(s_hi-s_lo,np.int64)   # rank for each sorted index
    T=len(tables)
    for bi in sel:
        s0,s1=bnd[bi],bnd[bi+1]
        idx=o[s0:s1]; c=codes[idx]
        g0,g1=s0//32,(s1-1)//32
        G=g1-g0+1
        cap=np.array([min(s1,32*(g+1))-max(s0,32*g) for g in range(g0,g1+1)])
        base=np.array([max(s0,32*g) for g in range(g0,g1+1)])
        col=(c[:,tables]&31).astype(np.int64)
        load=np.zeros((G,T,32),np.int32); fill=np.zeros(G,np.int32); seen=np.zeros((G,T,256),bool); val=c[:,tables].astype(np.int64)
        for r in range(len(idx)):      # arrival order inside the bucket
            add=np.zeros(G)
            for t in range(T): add+=(~seen[:,t,val[r,t]])*(2*load[:,t,col[r,t]]+1)
            add=add.astype(np.float64); add[fill>=cap]=1e9
            g=int(np.argmin(add))       # lowest group wins ties (what a lane-min reduction does)
            newpos[s0-s_lo+r]=base[g]+fill[g]; fill[g]+=1
            for t in range(T):
                if not seen[g,t,val[r,t]]:
                    seen[g,t,val[r,t]]=True; load[g,t,col[r,t]]+=1
    final=np.empty(s_hi-s_lo,np.int64); final[newpos-s_lo]=o[s_lo:s_hi]
    a=(s_lo+31)//32*32-s_lo; b=(s_hi-s_lo-a)//32*32
    gg=codes[final[a:a+b]].reshape(-1,32,m)
    p=group_passes(gg)
    base_p=group_passes(codes[o[s_lo+a:s_lo+a+b]].reshape(-1,32,m))
    print(label,"bits",bits[:6],"greedy",T,"tables: passes",p.mean(0).round(2),"sum %.2f"%p.sum(1).mean(),"| same sort without greedy %.2f"%base_p.sum(1).mean(),flush=True)
codes = np.load(sys.argv[1]) if len(sys.argv) > 1 else bench_codes()
o15=None
run(codes,[3,3,3,3,3,0,0,0],[5,6,7],600,"m8")
run(codes,[3,3,3,3,0,0,0,0],[4,5,6,7],150,"m8")
run(codes,[3,3,3,2,0,0,0,0],[3,4,5,6,7],80,"m8")
run(codes,[3,3,3,1,0,0,0,0],[3,4,5,6,7],40,"m8")
run(codes,[3,3,3,0,0,0,0,0],[3,4,5,6,7],20,"m8")
r16=rng.integers(0,256,(1_000_000,16),dtype=np.uint8)
run(r16,[3,3,3,3,3]+[0]*11,list(range(5,16)),600,"m16 random")
run(r16,[3,3,3,3]+[0]*12,list(range(4,16)),150,"m16 random")
run(r16,[3,3,3]+[0]*13,list(range(3,16)),20,"m16 random")
