"""Training phase clock at Deep1M shape (d = 96, m = 16: sub = 6 -- 8-byte gathers, 16 matrix-core units).  GPU box."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayuela_jl_amd as rq
import rayuela_jl_amd.synth as synth, rayuela_jl_amd.synth_torch as st
from rayuela_jl_amd import _lib
dev = torch.device("cuda", 0)
n, d, m, h = 1_000_000, 96, 16, 256
X = torch.cat([st.deep_like(250_000, d, seed=synth.SEED_BASE, row0=o, device=dev) for o in range(0, n, 250_000)], 0).cpu().numpy()
for name, fn in (("train_opq", lambda it: rq.train_opq(X, m, h, it, "natural", seed=7)), ("train_pq", lambda it: rq.train_pq(X, m, h, it, seed=7))):
    fn(3)
    fn(25)
    p = _lib.train_profile()
    rq.set_tuning("TRAIN_PROFILE", 1)
    fn(25)
    f = _lib.train_profile()
    rq.set_tuning("TRAIN_PROFILE", 0)
    it = max(1.0, f["iterations"])
    print(name, "ms/iter %.3f" % (p["loop_ms"] / max(1.0, p["iterations"])), {k: round(f[k] / it, 3) for k in ("qerror_ms", "gram_ms", "svd_ms", "rotate_ms", "update_centers_ms", "encode_ms", "reconstruct_ms", "converge_ms") if f[k] > 0},
          "init %.1f" % f["init_ms"], "ns_steps %.1f" % (p["ns_steps"] / max(1.0, p["iterations"])))
