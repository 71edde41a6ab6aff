#!/usr/bin/env python3
"""Encode kernels A/B on the bench's data (development aid): ms per 1e6 vectors and equality of the codes with the
f32-MFMA kernel (ENC_SPLIT=0), for the SIFT1M and Deep1M shapes."""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import rayuela_jl_amd as rq
import rayuela_jl_amd.synth as synth
import rayuela_jl_amd.synth_torch as st
from rayuela_jl_amd import device as rqd

dev = torch.device("cuda", 0)


def bench(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for kind in sys.argv[1:] or ["sift", "deep"]:
    n, h = 1_000_000, 256
    if kind == "sift":
        d, m = 128, 8
        X = torch.cat([st.sift_like(250_000, d, seed=synth.SEED_BASE, ncentres=65536, row0=o, device=dev) for o in range(0, n, 250_000)], 0)
    else:
        d, m = 96, 16
        X = torch.cat([st.deep_like(250_000, d, seed=synth.SEED_BASE, row0=o, device=dev) for o in range(0, n, 250_000)], 0)
    C = synth.codebooks(X[:20000].cpu().numpy(), m, h, seed=synth.SEED_CODEBOOK, iters=3, sample=20000)
    Ccat = torch.from_numpy(synth.cat_codebooks(C)).to(dev)
    out = torch.empty((n, m), dtype=torch.uint8, device=dev)
    rq.set_tuning("ENC_SPLIT", 0)
    ref = rqd.encode_pq(X, Ccat, m, h).clone()
    t0 = bench(lambda: rqd.encode_pq(X, Ccat, m, h, out=out))
    print("%s f32-MFMA kernel          %.4f ms" % (kind, t0))
    rq.set_tuning("ENC_SPLIT", 1)
    for w in (16, 12, 8):
        rq.set_tuning("ENC_SPLIT_WAVES", w)
        got = rqd.encode_pq(X, Ccat, m, h)
        torch.cuda.synchronize()
        diff = int((got != ref).sum())
        t1 = bench(lambda: rqd.encode_pq(X, Ccat, m, h, out=out))
        print("%s split kernel, %2d waves   %.4f ms   codes differing from the f32 kernel: %d of %d" % (kind, w, t1, diff, n * m))
    rq.set_tuning("ENC_SPLIT_WAVES", 0)
