#!/usr/bin/env python3
"""Scan timings with and without the bank-aware row order on bench-shaped data (development aid).
  raw0   rqd.linscan, SCAN_ORDER=0 (arrival order)
  raw1   rqd.linscan, default tuning (orders a scratch copy inside the call when nq >= 2048)
  ob     rqd.linscan over an OrderedBase made once (rq_dev_order_rows), + the time of making it
usage: python tools/order_perf.py [--n 1000000] [--nq 10000] [--ks 1,100,1000,10000] [--deep] [--uniform]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rayuela_jl_amd as rq  # noqa: E402
import rayuela_jl_amd.synth as synth  # noqa: E402
import rayuela_jl_amd.synth_torch as st  # noqa: E402
from rayuela_jl_amd import device as rqd  # noqa: E402
from rayuela_jl_amd import _lib  # noqa: E402


def bench(fn, iters=5, warm=8):
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
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--ks", default="1,100,1000,10000")
ap.add_argument("--uniform", action="store_true")
ap.add_argument("--deep", action="store_true")
ap.add_argument("--modes", default="raw0,raw1,ob")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--qseed", type=int, default=0, help="queries from another stream (seed) instead of rows beyond the base")
a = ap.parse_args()
dev = torch.device("cuda", 0)
d, m = (96, 16) if a.deep else (128, 8)
h, n, nq = 256, a.n, a.nq
gen = (lambda rows, row0: st.deep_like(rows, d, seed=synth.SEED_BASE, row0=row0, device=dev)) if a.deep else \
      (lambda rows, row0: st.sift_like(rows, d, seed=synth.SEED_BASE, ncentres=65536, row0=row0, device=dev))
Q = gen(nq, 3_000_000_000)
if a.qseed:
    Q = (st.deep_like(nq, d, seed=a.qseed, row0=0, device=dev) if a.deep else st.sift_like(nq, d, seed=a.qseed, ncentres=65536, row0=0, device=dev))
S = gen(20_000, 3_100_000_000)
C = synth.codebooks(S.cpu().numpy(), m, h, seed=synth.SEED_CODEBOOK, iters=5, sample=20000)
Ccat = torch.from_numpy(synth.cat_codebooks(C)).to(dev)
centers = torch.from_numpy(np.stack(C)).to(dev)
if a.uniform or n > 20_000_000:
    codes = rqd.synth_codes(n, m, seed=1234)
else:
    X = torch.cat([gen(min(250_000, n - o), o) for o in range(0, n, 250_000)], 0)
    codes = rqd.encode_pq(X, Ccat, m, h)
    del X
stats = bool(os.environ.get("RQ_SCAN_STATS"))
ref = {}
for mode in a.modes.split(","):
    base = codes
    rq.set_tuning("SCAN_ORDER", 0 if mode == "raw0" else 1)
    if mode == "ob":
        t_ord = bench(lambda: rqd.order_rows(codes), 3, 1)
        base = rqd.order_rows(codes)
        print("order_rows n=%d m=%d  %8.3f ms" % (n, m, t_ord), flush=True)
    for K in [int(x) for x in a.ks.split(",")]:
        out = (torch.empty((nq, K), dtype=torch.float32, device=dev), torch.empty((nq, K), dtype=torch.int32, device=dev))
        if stats:
            _lib.scan_stats()
        ms = bench(lambda: rqd.linscan(base, centers, Q, K, out=out), a.iters)
        extra = ""
        if stats:
            s = _lib.scan_stats()
            tot = sum(s[k] for k in ("lut", "sample", "stream", "final_cut", "sort_write")) or 1
            extra = "  fallbacks=%d cuts=%d items=%d | %s first_block_alive=%.2f%%" % (
                s["n_fallbacks"], s["n_cuts"], s["n_items"],
                " ".join("%s=%.1f" % (k, 100.0 * s[k] / tot) for k in ("lut", "sample", "stream", "cuts", "final_cut", "sort_write", "sample_rows", "sort_load", "sort_stages", "sort_out")),
                100.0 * s["first_block_pushed"] / max(1, s["first_block_rows"]))
        if K not in ref:
            ref[K] = (out[0].clone(), out[1].clone())
        else:
            extra += "  same=%s" % bool(torch.equal(out[0].view(torch.int32), ref[K][0].view(torch.int32)) and torch.equal(out[1], ref[K][1]))
        print("%-5s n=%d nq=%d m=%d K=%-5d %8.3f ms  %10.0f q/s%s" % (mode, n, nq, m, K, ms, nq / ms * 1e3, extra), flush=True)
rq.set_tuning("SCAN_ORDER", 1)
