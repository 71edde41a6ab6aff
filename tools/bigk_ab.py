#!/usr/bin/env python3
"""K > 1024 finish A/B on random tables: buckets from the distance map (SCAN_SS_MAP = 1) against sorted splitters (0), alternating.
usage: python tools/bigk_ab.py"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
import rayuela_jl_amd as rq
from rayuela_jl_amd import device as rqd
dev="cuda"
def bench(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
g = torch.Generator(device=dev).manual_seed(1)
for (n, m, sub, nq, K) in ((1_000_000, 16, 6, 1000, 4096), (1_000_000, 4, 8, 4096, 4096), (1_000_000, 8, 16, 4096, 4096), (500_000, 32, 4, 1000, 4096)):
    codes = rqd.synth_codes(n, m, seed=1234)
    centers = torch.randn((m, 256, sub), generator=g, device=dev) * 10
    queries = torch.randn((nq, m * sub), generator=g, device=dev) * 10
    out = (torch.empty((nq, K), dtype=torch.float32, device=dev), torch.empty((nq, K), dtype=torch.int32, device=dev))
    for mode in (1, 0, 1, 0):
        rq.set_tuning("SCAN_SS_MAP", mode)
        ms = bench(lambda: rqd.linscan(codes, centers, queries, K, out=out))
        print("n=%d m=%d nq=%d K=%d ss_map=%d %.4f ms" % (n, m, nq, K, mode, ms), flush=True)
rq.set_tuning("SCAN_SS_MAP", 1)
