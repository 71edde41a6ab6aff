"""Where does the wall time of a training call go?  (run on the GPU box)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayuela_jl_amd as rq
import rayuela_jl_amd.synth as synth, rayuela_jl_amd.synth_torch as st
from rayuela_jl_amd import _lib
dev = torch.device("cuda", 0)
n, d, m, h = 1_000_000, 128, 8, 256
X = torch.cat([st.sift_like(250_000, d, seed=synth.SEED_BASE, ncentres=65536, row0=o, device=dev) for o in range(0, n, 250_000)], 0).cpu().numpy()
for name, fn in (("train_opq", lambda it: rq.train_opq(X, m, h, it, "natural", seed=7)), ("train_pq", lambda it: rq.train_pq(X, m, h, it, seed=7))):
    fn(1)
    for it in (1, 25, 25):
        t0 = time.perf_counter(); fn(it); w = (time.perf_counter() - t0) * 1e3
        p = _lib.train_profile()
        print("%s niter=%2d wall %.1f ms  h2d %.1f init %.1f loop %.1f d2h %.1f  -> unaccounted %.1f" % (
            name, it, w, p["h2d_ms"], p["init_ms"], p["loop_ms"], p["d2h_ms"], w - p["h2d_ms"] - p["init_ms"] - p["loop_ms"] - p["d2h_ms"]))
