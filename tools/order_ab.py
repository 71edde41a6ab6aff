#!/usr/bin/env python3
"""A/B of base-row orders on the scan kernel (VERDICT r3 Next #1).  Bench-shaped data (bench.py's generators), the
codes permuted on the device with torch, rqd.linscan timed on each order.  Orders:
  arrival   as encoded
  lex       lexicographic by (b0, b1, ...)
  b15       counting-sort order by the top 3 bits of the first five code bytes (a 32-value window per byte = one LDS
            slot column each: conflict-free gathers), rows dealt so that the kernel's 32-lane groups are 32 consecutive
            sorted rows, and the 128-row tiles shuffled (any prefix of the base is a stratified sample)
usage: python tools/order_ab.py [--n 1000000] [--nq 10000] [--ks 1,100,1000,10000] [--uniform] [--deep]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rayuela_jl_amd as rq  # noqa: E402,F401
import rayuela_jl_amd.synth as synth  # noqa: E402
import rayuela_jl_amd.synth_torch as st  # noqa: E402
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


def order_key(codes, bits):
    key = torch.zeros(codes.shape[0], dtype=torch.int64, device=codes.device)
    for k, nb in enumerate(bits):
        if nb:
            key = (key << nb) | (codes[:, k].long() >> (8 - nb))
    return key


def deal(order, n, shuffle=True, rpt=2):
    """sorted rank s -> position: groups of 32 consecutive ranks become the kernel's lane groups (lane j of half h,
    sub-row r of a 128-row wave tile sits at 128w + 64h + rpt*j + r), full tiles shuffled by a Weyl permutation"""
    s = torch.arange(n, device=order.device)
    nfull = n // 128
    w, t = s // 128, s % 128
    h, u = t // 64, t % 64
    r, j = u // 32, u % 32
    if shuffle and nfull > 1:
        a = int(nfull * 0.6180339887) | 1
        while np.gcd(a, nfull) != 1:
            a += 2
        w = torch.where(w < nfull, (w * a) % nfull, w)
    pos = torch.where(s < nfull * 128, 128 * w + 64 * h + 2 * j + r, s)
    perm = torch.empty(n, dtype=torch.int64, device=order.device)
    perm[pos] = order          # position -> original row
    return perm


ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--ks", default="1,100,1000,10000")
ap.add_argument("--uniform", action="store_true")
ap.add_argument("--deep", action="store_true")
ap.add_argument("--orders", default="arrival,lex,b15,b15ns")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--check", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0)
d, m = (96, 16) if a.deep else (128, 8)
h, n, nq = 256, a.n, a.nq
gen = (lambda rows, row0: st.deep_like(rows, d, seed=synth.SEED_BASE, row0=row0, device=dev)) if a.deep else \
      (lambda rows, row0: st.sift_like(rows, d, seed=synth.SEED_BASE, ncentres=65536, row0=row0, device=dev))
Q = gen(nq, 3_000_000_000)
S = gen(20_000, 3_100_000_000)
C = synth.codebooks(S.cpu().numpy(), m, h, seed=synth.SEED_CODEBOOK, iters=5, sample=20000)
Ccat = torch.from_numpy(synth.cat_codebooks(C)).to(dev)
centers = torch.from_numpy(np.stack(C)).to(dev)
if a.uniform:
    codes = rqd.synth_codes(n, m, seed=1234)
else:
    X = torch.cat([gen(min(250_000, n - o), o) for o in range(0, n, 250_000)], 0)
    codes = rqd.encode_pq(X, Ccat, m, h)
    del X
ref = {}
for name in a.orders.split(","):
    if name == "arrival":
        perm = None
    elif name == "lex":
        perm = torch.argsort(order_key(codes, [8] * min(m, 7)), stable=True)
    elif name.startswith("b15"):
        o = torch.argsort(order_key(codes, [3, 3, 3, 3, 3] + [0] * (m - 5)), stable=True)
        perm = deal(o, n, shuffle=not name.endswith("ns"))
    elif name == "b15lin":      # bucket order without the dealing (the kernel's groups are every other row of 64)
        perm = torch.argsort(order_key(codes, [3, 3, 3, 3, 3] + [0] * (m - 5)), stable=True)
    else:
        raise SystemExit("unknown order " + name)
    cp = codes if perm is None else codes[perm].contiguous()
    for K in [int(x) for x in a.ks.split(",")]:
        out = (torch.empty((nq, K), dtype=torch.float32, device=dev), torch.empty((nq, K), dtype=torch.int32, device=dev))
        if os.environ.get("RQ_SCAN_STATS"):
            from rayuela_jl_amd import _lib
            _lib.scan_stats()
        ms = bench(lambda: rqd.linscan(cp, centers, Q, K, out=out), a.iters)
        extra = ""
        if os.environ.get("RQ_SCAN_STATS"):
            stt = _lib.scan_stats()
            extra = "  fallbacks=%d cuts=%d items=%d" % (stt["n_fallbacks"], stt["n_cuts"], stt["n_items"])
        if a.check:
            dd, ii = out[0].clone(), out[1].long()
            if perm is not None:
                ii = perm[ii]
            if K not in ref:
                ref[K] = (dd, ii)
                extra += "  (reference)"
            else:
                same_d = bool(torch.equal(dd.view(torch.int32), ref[K][0].view(torch.int32)))
                # ids: ties may be ordered differently when positions change; compare as (dist, id) sets per query
                k1 = torch.sort(dd.view(torch.int32).long() * (1 << 32) + ii, dim=1).values
                k0 = torch.sort(ref[K][0].view(torch.int32).long() * (1 << 32) + ref[K][1], dim=1).values
                extra += "  dists_equal=%s keysets_equal=%.6f" % (same_d, float((k1 == k0).float().mean()))
        print("%-8s n=%d nq=%d m=%d K=%-5d %8.3f ms  %10.0f q/s%s" % (name, n, nq, m, K, ms, nq / ms * 1e3, extra), flush=True)
