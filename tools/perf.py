#!/usr/bin/env python3
"""Quick kernel timings on resident synthetic data (development aid; bench.py is the contract).
usage: python tools/perf.py [scan] [encode] [rotate] [--n 1000000] [--nq 10000] [--m 8] [--ks 1,100,1000]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rayuela_jl_amd as rq  # noqa: E402
from rayuela_jl_amd import device as rqd  # noqa: E402


def bench(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


ap = argparse.ArgumentParser()
ap.add_argument("what", nargs="*", default=["scan", "encode", "rotate"])
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--m", type=int, default=8)
ap.add_argument("--d", type=int, default=128)
ap.add_argument("--ks", default="1,100,1000")
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
n, nq, m, d = a.n, a.nq, a.m, a.d
sub = d // m
if "scan" in a.what:
    codes = rqd.synth_codes(n, m, seed=1234)
    centers = torch.randn((m, 256, sub), generator=g, device=dev) * 10
    queries = torch.randn((nq, d), generator=g, device=dev) * 10
    for K in [int(x) for x in a.ks.split(",")]:
        out = (torch.empty((nq, K), dtype=torch.float32, device=dev), torch.empty((nq, K), dtype=torch.int32, device=dev))
        ms = bench(lambda: rqd.linscan(codes, centers, queries, K, out=out), a.iters)
        if os.environ.get("RQ_SCAN_STATS"):
            from rayuela_jl_amd import _lib
            st = _lib.scan_stats()
            tot = sum(st[k] for k in ("lut", "sample", "stream", "final_cut", "sort_write")) or 1
            print("   phases(%%): " + " ".join("%s=%.1f" % (k, 100.0 * st[k] / tot) for k in ("lut", "sample", "stream", "cuts", "final_cut", "sort_write")) + " n_cuts=%d n_fallbacks=%d sample_rows=%.1f%% sort_load=%.1f%% sort_stages=%.1f%% sort_out=%.1f%% items=%d filtered=%d first_block_alive=%.2f%%" % (st["n_cuts"], st["n_fallbacks"], 100.0 * st["sample_rows"] / tot, 100.0 * st["sort_load"] / tot, 100.0 * st["sort_stages"] / tot, 100.0 * st["sort_out"] / tot, st["n_items"], st["n_items_filtered"], 100.0 * st["first_block_pushed"] / max(1, st["first_block_rows"])))
        print("scan   n=%d nq=%d m=%d K=%-5d %8.3f ms  %10.0f q/s  %7.1f GB/s-alg" % (n, nq, m, K, ms, nq / ms * 1e3, nq * n * m / ms / 1e6))
if "aq" in a.what:   # linscan_lsq / linscan_cq: full-dimensional codebooks, LSQ adds the row norms
    codes = rqd.synth_codes(n, m, seed=1234)
    cb = torch.randn((m * 256, d), generator=g, device=dev)
    queries = torch.randn((nq, d), generator=g, device=dev)
    norms = torch.rand((n,), generator=g, device=dev) * 100
    for K in [int(x) for x in a.ks.split(",")]:
        out = (torch.empty((nq, K), dtype=torch.float32, device=dev), torch.empty((nq, K), dtype=torch.int32, device=dev))
        for name, nr in (("lsq", norms), ("cq", None)):
            ms = bench(lambda: rqd.linscan_aq(codes, cb, queries, K, dbnorms=nr, out=out), a.iters)
            print("%-6s n=%d nq=%d m=%d d=%d K=%-5d %8.3f ms  %10.0f q/s" % (name, n, nq, m, d, K, ms, nq / ms * 1e3))
if "encode" in a.what:
    X = torch.randint(0, 200, (n, d), generator=g, device=dev).float()
    C = torch.randint(0, 200, (256 * d,), generator=g, device=dev).float()
    out = torch.empty((n, m), dtype=torch.uint8, device=dev)
    for w in (8, 16):
        rq.set_tuning("ENC_WAVES", w)
        ms = bench(lambda: rqd.encode_pq(X, C, m, 256, out=out), a.iters)
        print("encode n=%d d=%d m=%d waves=%d %8.3f ms  %12.0f vec/s  %6.1f TF" % (n, d, m, w, ms, n / ms * 1e3, 2.0 * d * 256 * n / ms / 1e9))
    rq.set_tuning("ENC_WAVES", 8)
if "rotate" in a.what:
    X = torch.randn((n, d), generator=g, device=dev)
    R = torch.randn((d, d), generator=g, device=dev)
    out = torch.empty_like(X)
    ms = bench(lambda: rqd.rotate_T(R, X, out=out), a.iters)
    print("rotate n=%d d=%d %8.3f ms  %6.1f TF  %7.1f GB/s" % (n, d, ms, 2.0 * d * d * n / ms / 1e9, 8.0 * d * n / ms / 1e6))

if "train" in a.what:
    import time
    import numpy as np
    X = torch.randint(0, 200, (n, d), generator=g, device=dev).float()
    C = torch.randint(0, 200, (256 * d,), generator=g, device=dev).float()
    R = torch.from_numpy(np.linalg.qr(np.random.default_rng(0).standard_normal((d, d)))[0].astype(np.float32)).to(dev)
    codes = rqd.encode_pq(X, C, m, 256)
    CB = rqd.reconstruct(codes, C, d, 256)
    for name, fn in (("update_centers", lambda: rqd.update_centers(C, X, codes, m, 256)),
                     ("reconstruct", lambda: rqd.reconstruct(codes, C, d, 256, out=CB)),
                     ("qerror", lambda: rqd.qerror(X, CB)),
                     ("gram X'CB", lambda: rqd.gram(X, CB))):
        print("train  %-15s n=%d d=%d  %8.3f ms" % (name, n, d, bench(fn, a.iters)))
    t0 = time.perf_counter()
    U, S, Vt = np.linalg.svd(rqd.gram(X, CB).cpu().numpy().astype(np.float64))
    print("train  host SVD %dx%d      %8.3f ms" % (d, d, (time.perf_counter() - t0) * 1e3))
