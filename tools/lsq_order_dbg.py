import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import rayuela_jl_amd as rq, rayuela_jl_amd.synth as synth
from rayuela_jl_amd import _lib
m, K, n, nq, h, d = 8, 1000, 1_000_000, 10_000, 256, 128
X = synth.sift_like(n, d, seed=synth.SEED_BASE); Q = synth.sift_like(nq, d, seed=synth.SEED_QUERY)
S = synth.sift_like(20_000, d, seed=synth.SEED_BASE, row0=3_100_000_000)
C = synth.codebooks(S, m, h, seed=synth.SEED_CODEBOOK, iters=5, sample=20000)
B = rq.quantize_pq_u8(X, C); sub = d // m
cb = np.zeros((m * h, d), dtype=np.float32)
for i in range(m): cb[i*h:(i+1)*h, i*sub:(i+1)*sub] = C[i]
Xhat = np.concatenate([C[i][B[:, i]] for i in range(m)], axis=1)
norms = (Xhat.astype(np.float64) ** 2).sum(1).astype(np.float32)
Cl = [cb[i*h:(i+1)*h] for i in range(m)]; R = np.eye(d, dtype=np.float32)
for order in (1, 0):
    rq.set_tuning("INDEX_ORDER", order)
    with rq.LsqIndex(B, Cl, norms) as ix:
        ix.search(Q, R, K)
        rq.set_tuning("SCAN_STATS", 1)
        ix.search(Q, R, K)
        st = _lib.scan_stats(); rq.set_tuning("SCAN_STATS", 0)
        best = 1e9
        for _ in range(3):
            ix.search(Q, R, K); best = min(best, rq.last_timing()["kernel_ms"])
        print("order", order, "kernel %.2f ms" % best, {k: st[k] for k in ("n_items", "n_items_filtered", "n_fallbacks", "n_cuts")}, "alive %.1f%%" % (100.0*st["first_block_pushed"]/max(1,st["first_block_rows"])))
rq.set_tuning("INDEX_ORDER", 1)
for gran in (128, 256, 0):
    rq.set_tuning("ORDER_GRAN", gran)
    with rq.LsqIndex(B, Cl, norms) as ix:
        ix.search(Q, R, K)
        rq.set_tuning("SCAN_STATS", 1); ix.search(Q, R, K); st = _lib.scan_stats(); rq.set_tuning("SCAN_STATS", 0)
        best = min(( (ix.search(Q, R, K), rq.last_timing()["kernel_ms"])[1] for _ in range(3)))
        print("gran", gran, "kernel %.2f ms" % best, {k: st[k] for k in ("n_items", "n_fallbacks", "n_cuts")})
rq.set_tuning("ORDER_GRAN", 0)
Cpq = [C[i] for i in range(m)]
ixp = rq.Index(Cpq, d); ixp.set_codes(B)
ixp.search(Q, K); rq.set_tuning("SCAN_STATS", 1); ixp.search(Q, K); st = _lib.scan_stats(); rq.set_tuning("SCAN_STATS", 0)
print("PQ index on the same codes:", {k: st[k] for k in ("n_items", "n_fallbacks", "n_cuts")})
