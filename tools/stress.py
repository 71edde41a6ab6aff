#!/usr/bin/env python3
"""Randomised stress of the scan and the encode against the oracle, beyond tests/test_gpu_fuzz.py: larger
query batches (mixed whole / sliced plans), K up to 20000 (sample-sort finish and merge), wide sub-spaces,
RVQ.  usage: python tools/stress.py [--rounds 150] [--seed 0]   (prints every failing configuration)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rayuela_jl_amd as rq  # noqa: E402
import rayuela_jl_amd.synth as synth  # noqa: E402
from oracle import oracle  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=150)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
fails = 0
t0 = time.time()
for r in range(a.rounds):
    rng = np.random.default_rng(a.seed * 100003 + r)
    kind = r % 5
    if kind == 3:      # scan shapes that run the integer pre-filter (rows >= 64 k, m in {8, 16}) on hostile tables
        m = int(rng.choice([8, 8, 16]))
        sub = int(rng.choice([1, 2, 6]))
        n = int(rng.choice([70000, 300000, 1000000]))
        nq = int(rng.choice([3, 8, 40]))
        K = int(rng.choice([1, 10, 100, 1000]))
        style = int(rng.integers(0, 7))
        centers = rng.standard_normal((m, 256, sub)).astype(np.float32)
        queries = rng.standard_normal((nq, m * sub)).astype(np.float32)
        codes = synth.random_codes(n, m, seed=r)
        if style == 1:      # all table entries equal per sub-quantizer: no contrast at all
            centers[:] = centers[:, :1, :]
        elif style == 2:    # one sub-quantizer dominates the distance
            centers[0] *= 1000.0
        elif style == 3:    # integer tables: massive distance ties
            centers = rng.integers(0, 3, (m, 256, sub)).astype(np.float32)
            queries = rng.integers(0, 3, (nq, m * sub)).astype(np.float32)
        elif style == 4:    # few distinct rows: the same codes again and again
            codes = codes[rng.integers(0, 50, n)]
        elif style == 5:    # large common offset: tau ~ sum of the table minima (cancellation in the filter's range)
            queries += 1000.0
        elif style == 6:    # clustered: a small share of rows is close, the rest far (high-contrast tables)
            centers[:, 8:, :] += 30.0
            queries *= 0.1
        d0, i0 = oracle.linscan_aqd_query(codes, centers, queries, K)
        d1, i1 = rq.linscan_aqd_query(codes, centers, queries, K)
        ok = np.array_equal(i0, i1) and np.array_equal(bits(d0), bits(d1))
        desc = "filter-scan m=%d sub=%d n=%d nq=%d K=%d style=%d" % (m, sub, n, nq, K, style)
    elif kind == 4:      # LSQ / CQ scans at sizes that run their pre-filters, hostile norms and codebooks
        m = int(rng.choice([8, 8, 16, 7, 12]))
        d = int(rng.choice([16, 24, 40]))
        n = int(rng.choice([70000, 200000, 600000]))
        nq = int(rng.choice([3, 8, 24]))
        K = int(rng.choice([1, 10, 100, 1000]))
        style = int(rng.integers(0, 8))
        h = 256
        cb = rng.standard_normal((m * h, d)).astype(np.float32)
        queries = rng.standard_normal((nq, d)).astype(np.float32)
        codes = synth.random_codes(n, m, seed=r)
        if style in (1, 7):      # near-orthogonal codebooks
            w = max(1, d // m)
            cb[:] = 0
            for i in range(m):
                a0 = (i * w) % (d - w + 1)
                cb[i * h:(i + 1) * h, a0:a0 + w] = rng.standard_normal((h, w)).astype(np.float32) * 3
        elif style == 2:         # common offset: large cross terms
            cb += rng.standard_normal((1, d)).astype(np.float32) * 5
        elif style == 5:
            codes = codes[rng.integers(0, 40, n)]
        xh = np.zeros((n, d))
        for i in range(m):
            xh += cb[i * h + codes[:, i].astype(np.int64)]
        norms = (xh ** 2).sum(1).astype(np.float32)
        if style == 3:
            norms[:] = float(rng.standard_normal() * 10)
        elif style == 4:
            norms = (rng.standard_normal(n) * 1e4).astype(np.float32)
        elif style == 6:
            norms = -norms
        C = [cb[i * h:(i + 1) * h] for i in range(m)]
        if style == 7:           # CQ: no norms, non-negative full-dimensional tables
            d0, i0 = oracle.linscan_cq(codes, cb, queries, K)
            d1, i1 = rq.linscan_cq(codes, queries, C, K)
        else:
            d0, i0 = oracle.linscan_lsq(codes, cb, queries, norms, K)
            d1, i1 = rq.linscan_lsq(codes, queries, C, norms, np.eye(d, dtype=np.float32), K)
        ok = np.array_equal(i0.astype(np.int64), i1.astype(np.int64)) and np.array_equal(bits(d0), bits(d1))
        desc = "aq-scan m=%d d=%d n=%d nq=%d K=%d style=%d" % (m, d, n, nq, K, style)
    elif kind == 0:      # scan
        m = int(rng.choice([2, 4, 8, 8, 8, 16, 32, 64, 5, 11]))
        sub = int(rng.choice([1, 2, 4, 8]))
        n = int(rng.choice([3000, 40000, 90000, 250000]))
        nq = int(rng.choice([1, 7, 64, 500, 1100, 4100, 4500, 5200]))
        if nq > 2000:
            n = min(n, 40000)
        K = int(min(n, rng.choice([1, 10, 100, 1000, 1024, 1025, 2500, 5000, 20000])))
        ties = rng.random() < 0.4
        if ties:
            centers = rng.integers(0, 3, (m, 256, sub)).astype(np.float32)
            queries = rng.integers(0, 3, (nq, m * sub)).astype(np.float32)
            codes = (synth.random_codes(n, m, seed=r) % 5).astype(np.uint8)
        else:
            centers = rng.standard_normal((m, 256, sub)).astype(np.float32)
            queries = rng.standard_normal((nq, m * sub)).astype(np.float32)
            codes = synth.random_codes(n, m, seed=r)
        sl = int(rng.choice([0, 0, 0, 2, 5]))
        rq.set_tuning("SCAN_SLICES", sl)
        d0, i0 = oracle.linscan_aqd_query(codes, centers, queries, K)
        d1, i1 = rq.linscan_aqd_query(codes, centers, queries, K)
        rq.set_tuning("SCAN_SLICES", 0)
        ok = np.array_equal(i0, i1) and np.array_equal(bits(d0), bits(d1))
        desc = "scan m=%d sub=%d n=%d nq=%d K=%d ties=%d slices=%d" % (m, sub, n, nq, K, ties, sl)
    elif kind == 1:    # encode, including wide sub-spaces
        m = int(rng.choice([1, 2, 3, 4, 8, 16]))
        sub = int(rng.choice([1, 4, 16, 33, 48, 64, 65, 96, 100, 128, 150]))
        extra = int(rng.choice([0, 0, 1])) if m > 1 else 0
        d = m * sub + extra
        h = int(rng.choice([2, 31, 64, 100, 256]))
        n = int(rng.choice([1, 33, 1000, 4099]))
        X = (rng.standard_normal((n, d)) * 5).astype(np.float32)
        off = synth.splitarray(d, m)
        C = [np.round(rng.standard_normal((h, int(off[i + 1] - off[i]))) * 5).astype(np.float32) for i in range(m)]
        c0 = oracle.encode_pq(X, synth.cat_codebooks(C), m, h)
        c1 = rq.quantize_pq_u8(X, C)
        ok = np.array_equal(c0, c1)
        desc = "encode m=%d d=%d h=%d n=%d" % (m, d, h, n)
    else:              # RVQ
        d = int(rng.choice([8, 30, 64, 96, 128, 200]))
        m = int(rng.choice([1, 2, 5]))
        h = int(rng.choice([16, 100, 256]))
        n = int(rng.choice([40, 1000, 6000]))
        X = synth.deep_like(n, d, seed=r + 1)
        C = synth.rvq_codebooks(X, m, h, seed=r + 2, iters=1, sample=min(n, 1024))
        c0, cnt0, x0 = oracle.encode_rvq(X, C, with_extras=True)
        c1, cnt1, x1 = rq.quantize_rvq_u8(X, [C[i] for i in range(m)], with_extras=True)
        ok = np.array_equal(c0, c1) and np.array_equal(cnt0, cnt1) and np.array_equal(bits(x0), bits(x1))
        desc = "rvq d=%d m=%d h=%d n=%d" % (d, m, h, n)
    if not ok:
        fails += 1
        print("FAIL", desc, flush=True)
print("stress: %d rounds, %d failures, %.1f s" % (a.rounds, fails, time.time() - t0))
sys.exit(1 if fails else 0)
