"""Timing of the device polar factor on the Gram matrix of the SIFT1M-shape bench data (run on the GPU box)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayuela_jl_amd as rq
import rayuela_jl_amd.synth as synth, rayuela_jl_amd.synth_torch as st
from rayuela_jl_amd import device as rqd, _lib
dev = torch.device("cuda", 0)
d, m, h, n = 128, 8, 256, 200_000
X = st.sift_like(n, d, seed=synth.SEED_BASE, ncentres=65536, row0=0, device=dev)
C = synth.codebooks(X[:20000].cpu().numpy(), m, h, seed=synth.SEED_CODEBOOK, iters=3, sample=20000)
Ccat = torch.from_numpy(synth.cat_codebooks(C)).to(dev)
codes = rqd.encode_pq(X, Ccat, m, h)
CB = rqd.reconstruct(codes, Ccat, d, h)
G = rqd.gram(X, CB)
s = np.linalg.svd(G.cpu().numpy().astype(np.float64), compute_uv=False)
print("G: sigma max %.3e min %.3e  kappa %.1f  fro/sigma_max %.2f" % (s[0], s[-1], s[0] / s[-1], np.sqrt((s**2).sum()) / s[0]))
for l0 in (1000, 10000, 100, 100000, 300000):
    rq.set_tuning("TRAIN_NS_L0_MICRO", l0)
    for method in (0, 1):
        R, ok, steps = rqd.polar_factor(G, method)
        t0 = time.perf_counter()
        for _ in range(20):
            R, ok, steps = rqd.polar_factor(G, method)
        dt = (time.perf_counter() - t0) / 20
        print("l0 %7d method %d ok %s steps %d  %.3f ms per call (incl. alloc + sync)" % (l0, method, ok, steps, dt * 1e3))
rq.set_tuning("TRAIN_NS_L0_MICRO", 1000)
