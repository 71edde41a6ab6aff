#!/usr/bin/env python3
"""Round-5 finish A/B on random tables (random codes, Gaussian codebooks and queries): SCAN_BUCKET_FINISH = 1 / 0 alternating, with the
phase clock of the first call; shapes where the bucket finish first LOST (one item per workgroup, short slices).  usage: python tools/finish_ab_random.py"""
import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
import rayuela_jl_amd as rq
from rayuela_jl_amd import device as rqd, _lib
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
for (n, m, sub, nq, K) in ((1_000_000, 8, 16, 4096, 1000), (1_000_000, 4, 8, 4096, 1000), (500_000, 32, 4, 64, 1000), (1_000_000, 8, 16, 64, 1000)):
    codes = rqd.synth_codes(n, m, seed=1234)
    centers = torch.randn((m, 256, sub), generator=g, device=dev) * 10
    queries = torch.randn((nq, m * sub), generator=g, device=dev) * 10
    out = (torch.empty((nq, K), dtype=torch.float32, device=dev), torch.empty((nq, K), dtype=torch.int32, device=dev))
    for mode in (1, 0, 1, 0):
        rq.set_tuning("SCAN_BUCKET_FINISH", mode)
        rq.set_tuning("SCAN_STATS", 1); _lib.scan_stats()
        rqd.linscan(codes, centers, queries, K, out=out); torch.cuda.synchronize()
        s = _lib.scan_stats(); rq.set_tuning("SCAN_STATS", 0)
        ms = bench(lambda: rqd.linscan(codes, centers, queries, K, out=out))
        tot = sum(s[k] for k in ("lut", "sample", "stream", "final_cut", "sort_write")) or 1
        print("n=%d m=%d nq=%d K=%d mode=%d %.4f ms | final_cut=%.1f sort_write=%.1f (load %.1f stages %.1f out %.1f) items=%d" % (n, m, nq, K, mode, ms, 100*s["final_cut"]/tot, 100*s["sort_write"]/tot, 100*s["sort_load"]/tot, 100*s["sort_stages"]/tot, 100*s["sort_out"]/tot, s["n_items"]), flush=True)
rq.set_tuning("SCAN_BUCKET_FINISH", 1)
