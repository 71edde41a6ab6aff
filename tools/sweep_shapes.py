#!/usr/bin/env python3
"""ms per resident linscan call over 80 shapes (m = 4 ... 32, 8 ... 4096 queries, k = 10 ... 4096; random tables).  Run it twice on one box,
once with RAYUELA_HIP_LIB pointing at another build, to compare builds (profiles/r5_shape_sweep.md).  usage: python tools/sweep_shapes.py"""
import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
import rayuela_jl_amd as rq
from rayuela_jl_amd import device as rqd
dev = "cuda"
def bench(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
g = torch.Generator(device=dev).manual_seed(1)
for (n, m, sub) in ((1_000_000, 8, 16), (1_000_000, 16, 6), (1_000_000, 4, 8), (500_000, 32, 4), (4_000_000, 8, 16)):
    codes = rqd.synth_codes(n, m, seed=1234)
    centers = torch.randn((m, 256, sub), generator=g, device=dev) * 10
    for nq in (8, 64, 1000, 4096):
        queries = torch.randn((nq, m * sub), generator=g, device=dev) * 10
        for K in (10, 100, 1000, 4096):
            if K * 16 > n: continue
            out = (torch.empty((nq, K), dtype=torch.float32, device=dev), torch.empty((nq, K), dtype=torch.int32, device=dev))
            ms = bench(lambda: rqd.linscan(codes, centers, queries, K, out=out), 10 if nq >= 1000 else 30)
            print("n=%d m=%d nq=%d K=%d  %.4f" % (n, m, nq, K, ms), flush=True)
