import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import rayuela_jl_amd as rq
from rayuela_jl_amd import _lib, device as rqd
from oracle import oracle
rng = np.random.default_rng(1)
for (n, m, sub, nq, K) in ((200_000, 8, 16, 40, 100), (50_001, 8, 16, 13, 100), (200_000, 8, 16, 40, 1000), (70_000, 8, 4, 17, 4096)):
    centers = rng.standard_normal((m, 256, sub)).astype(np.float32)
    queries = rng.standard_normal((nq, m * sub)).astype(np.float32)
    codes = rng.integers(0, 256, (n, m), dtype=np.uint8)
    d0, i0 = oracle.linscan_aqd_query(codes, centers, queries, K)
    rq.set_tuning("SCAN_STATS", 1); _lib.scan_stats()
    d1, i1 = rq.linscan_aqd_query(codes, centers, queries, K)
    st = _lib.scan_stats(); rq.set_tuning("SCAN_STATS", 0)
    ok = np.array_equal(i0, i1) and np.array_equal(d0.view(np.uint32), d1.view(np.uint32))
    print((n, m, sub, nq, K), "OK" if ok else "MISMATCH rows with diff: %d" % int((i0 != i1).any(1).sum()), _lib.scan_plan(n, nq, m, m*sub, K),
          {k: st[k] for k in ("n_items", "n_items_filtered", "n_fallbacks", "first_block_pushed", "first_block_rows")}, flush=True)
