#!/usr/bin/env python3
"""PCIe-inclusive timing of the host-pointer entry points (what a Julia ccall pays)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rayuela_jl_amd as rq
import rayuela_jl_amd.synth as synth
n, d, m, h, nq, K = 1_000_000, 128, 8, 256, 10_000, 1000
rng = np.random.default_rng(0)
X = rng.integers(0, 200, (n, d)).astype(np.float32)
C = [rng.integers(0, 200, (h, d // m)).astype(np.float32) for _ in range(m)]
Q = rng.integers(0, 200, (nq, d)).astype(np.float32)
for it in range(3):
    t = time.perf_counter(); B = rq.quantize_pq_u8(X, C); dt = time.perf_counter() - t
    print("quantize_pq host->host  %.1f ms  %s  -> %.2e vec/s" % (dt * 1e3, {k: round(v, 1) for k, v in rq.last_timing().items()}, n / dt))
with rq.Dataset(X) as ds:
    for it in range(3):
        t = time.perf_counter(); B2 = ds.quantize(C, one_based=False); dt = time.perf_counter() - t
        print("resident dataset encode %.1f ms  -> %.2e vec/s" % (dt * 1e3, n / dt))
    assert np.array_equal(B, B2)
rq.set_tuning("HOST_OVERLAP", 0)
for it in range(2):
    t = time.perf_counter(); B = rq.quantize_pq_u8(X, C); dt = time.perf_counter() - t
    print("quantize_pq host->host, no overlap  %.1f ms" % (dt * 1e3))
rq.set_tuning("HOST_OVERLAP", 1)
for it in range(3):
    t = time.perf_counter(); D, I = rq.linscan_pq(B, Q, C, 8 * m, K); dt = time.perf_counter() - t
    print("linscan_pq  host->host  %.1f ms  %s  -> %.2e q/s" % (dt * 1e3, {k: round(v, 1) for k, v in rq.last_timing().items()}, nq / dt))
