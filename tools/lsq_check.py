#!/usr/bin/env python3
"""linscan_lsq on realistic tables: a PQ quantizer written as an additive one (each sub-codebook zero-padded to the full
dimension, norms = |x_hat|^2), so the LSQ scan sees clustered data.  Checks the GPU answer against the CPU oracle on a
query sample and times the scan with and without the LSQ pre-filter.   usage: python tools/lsq_check.py [m] [k]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rayuela_jl_amd as rq  # noqa: E402
import rayuela_jl_amd.synth as synth  # noqa: E402
from oracle import oracle  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n, nq, h = 1_000_000, 10_000, 256
d = 128 if m == 8 else 96
gen = synth.sift_like if m == 8 else synth.deep_like
X = gen(n, d, seed=synth.SEED_BASE)
Q = gen(nq, d, seed=synth.SEED_QUERY)
S = gen(20_000, d, seed=synth.SEED_BASE, row0=3_100_000_000)
C = synth.codebooks(S, m, h, seed=synth.SEED_CODEBOOK, iters=5, sample=20000)
B = rq.quantize_pq_u8(X, C)
sub = d // m
cb = np.zeros((m * h, d), dtype=np.float32)
for i in range(m):
    cb[i * h:(i + 1) * h, i * sub:(i + 1) * sub] = C[i]
Xhat = np.concatenate([C[i][B[:, i]] for i in range(m)], axis=1)
norms = (Xhat.astype(np.float64) ** 2).sum(1).astype(np.float32)
Cl = [cb[i * h:(i + 1) * h] for i in range(m)]
R = np.eye(d, dtype=np.float32)
use_ref = oracle.ref_aq_available()      # the compiled reference (deps/src/linscan_aqd_pairwise_byte.cpp) when it travelled along
d0, i0 = oracle.linscan_lsq(B, cb, Q[:64], norms, K, use_ref=use_ref)
print("checker: %s" % ("compiled reference" if use_ref else "oracle restatement"))
for flt in (1, 0):
    rq.set_tuning("SCAN_FILTER_LSQ", flt)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        D, I = rq.linscan_lsq(B, Q, Cl, norms, R, K)
        best = min(best, rq.last_timing()["kernel_ms"])
    same = np.array_equal(I[:64].astype(np.int64), i0.astype(np.int64)) and np.array_equal(D[:64].view(np.uint32), d0.view(np.uint32))
    print("m=%d k=%d filter=%d: kernel %.2f ms, same as the checker on 64 queries: %s" % (m, K, flt, best, same), flush=True)
rq.set_tuning("SCAN_FILTER_LSQ", 1)
# prepared base (rq_lsq_prepare): codes, norms, codebooks, the filter's O(n) pass and the bank-aware row order once
with rq.LsqIndex(B, Cl, norms) as ix:
    best = 1e9
    for _ in range(4):
        D, I = ix.search(Q, R, K)
        best = min(best, rq.last_timing()["kernel_ms"])
    same = np.array_equal(I[:64].astype(np.int64), i0.astype(np.int64)) and np.array_equal(D[:64].view(np.uint32), d0.view(np.uint32))
    print("m=%d k=%d prepared base (LsqIndex): kernel %.2f ms, same as the checker on 64 queries: %s" % (m, K, best, same), flush=True)

# resident timing + the kernel's own counters
import torch  # noqa: E402
from rayuela_jl_amd import device as rqd, _lib  # noqa: E402
Bd, cbd, Qd, nd = (torch.from_numpy(a).cuda() for a in (B, cb, Q, norms))
for flt in (1, 0):
    rq.set_tuning("SCAN_FILTER_LSQ", flt)
    rq.set_tuning("SCAN_STATS", 1)
    for _ in range(2):
        out = rqd.linscan_aq(Bd, cbd, Qd, K, dbnorms=nd)
    torch.cuda.synchronize()
    st = _lib.scan_stats()
    rq.set_tuning("SCAN_STATS", 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        out = rqd.linscan_aq(Bd, cbd, Qd, K, dbnorms=nd)
    e1.record()
    torch.cuda.synchronize()
    tot = sum(st[k] for k in ("lut", "sample", "stream", "final_cut", "sort_write"))
    print("resident m=%d k=%d filter=%d: %.2f ms  items=%d filtered=%d first-block alive=%.1f%% fallbacks=%d cuts=%d  %s" % (
        m, K, flt, e0.elapsed_time(e1) / 5, st["n_items"], st["n_items_filtered"],
        100.0 * st["first_block_pushed"] / max(1, st["first_block_rows"]), st["n_fallbacks"], st["n_cuts"],
        {k: round(100.0 * st[k] / tot, 1) for k in ("lut", "sample", "stream", "cuts", "final_cut", "sort_write")}), flush=True)
rq.set_tuning("SCAN_FILTER_LSQ", 1)

# the same codebooks as a CQ scan (non-negative full-dimensional tables, no norms): does the threshold sample hold there?
rq.set_tuning("SCAN_STATS", 1)
for _ in range(2):
    out = rqd.linscan_aq(Bd, cbd, Qd, K)
torch.cuda.synchronize()
st = _lib.scan_stats()
rq.set_tuning("SCAN_STATS", 0)
print("CQ on the same codebooks: items=%d filtered=%d first-block alive=%.1f%% fallbacks=%d" % (
    st["n_items"], st["n_items_filtered"], 100.0 * st["first_block_pushed"] / max(1, st["first_block_rows"]), st["n_fallbacks"]))
# and LSQ with every norm shifted so that all distances are positive
shift = float(np.abs(Q).max() * np.abs(cb).max() * d * 2 * m)
nd2 = nd + shift
rq.set_tuning("SCAN_STATS", 1)
rq.set_tuning("SCAN_FILTER_LSQ", 0)
for _ in range(2):
    out = rqd.linscan_aq(Bd, cbd, Qd, K, dbnorms=nd2)
torch.cuda.synchronize()
st = _lib.scan_stats()
rq.set_tuning("SCAN_STATS", 0)
rq.set_tuning("SCAN_FILTER_LSQ", 1)
print("LSQ, norms + %.3g (all distances positive): items=%d fallbacks=%d" % (shift, st["n_items"], st["n_fallbacks"]))

# PQ scan of the very same codes and queries
cen = torch.from_numpy(np.stack(C)).cuda()
rq.set_tuning("SCAN_STATS", 1)
for _ in range(2):
    out = rqd.linscan(Bd, cen, Qd, K)
torch.cuda.synchronize()
st = _lib.scan_stats()
rq.set_tuning("SCAN_STATS", 0)
print("PQ scan of the same data: items=%d fallbacks=%d cuts=%d" % (st["n_items"], st["n_fallbacks"], st["n_cuts"]))
dd = out[0][:, :K].cpu().numpy()
print("distinct distance values among the top-%d of query 0 / 1 / 2: %d %d %d" % (K, len(np.unique(dd[0])), len(np.unique(dd[1])), len(np.unique(dd[2]))))
outl = rqd.linscan_aq(Bd, cbd, Qd, K, dbnorms=nd)
dl = outl[0][:, :K].cpu().numpy()
print("LSQ: distinct values: %d %d %d" % (len(np.unique(dl[0])), len(np.unique(dl[1])), len(np.unique(dl[2]))))
