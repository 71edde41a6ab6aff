#!/usr/bin/env python3
"""Encode time against the row count (development aid): the intercept is the per-launch cost (prologue: LDS tables)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayuela_jl_amd as rq
import rayuela_jl_amd.synth as synth
import rayuela_jl_amd.synth_torch as st
from rayuela_jl_amd import device as rqd
dev = torch.device("cuda", 0)
d, m, h = 128, 8, 256
X = torch.cat([st.sift_like(250_000, d, seed=synth.SEED_BASE, ncentres=65536, row0=o, device=dev) for o in range(0, 1_000_000, 250_000)], 0)
C = synth.codebooks(X[:20000].cpu().numpy(), m, h, seed=synth.SEED_CODEBOOK, iters=3, sample=20000)
Ccat = torch.from_numpy(synth.cat_codebooks(C)).to(dev)
for n in (8192, 65536, 262144, 1_000_000):
    out = torch.empty((n, m), dtype=torch.uint8, device=dev)
    Xn = X[:n]
    for _ in range(3): rqd.encode_pq(Xn, Ccat, m, h, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): rqd.encode_pq(Xn, Ccat, m, h, out=out)
    e1.record(); torch.cuda.synchronize()
    print("n=%8d  %.4f ms" % (n, e0.elapsed_time(e1) / 20), flush=True)
