"""Cost of the query-chunk pipeline of rq_index_search on ONE GPU (logical shards): kernel_ms of the search at
SIFT1M shape for 1/2/4/8 chunks.  On one device nothing overlaps, so this is the pure overhead of splitting the scan."""
import sys
import numpy as np
sys.path.insert(0, ".")
import rayuela_jl_amd as rq
from rayuela_jl_amd import _lib, synth

n, nq, m, d, k = 1_000_000, 10_000, 8, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 1000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(0)
C = [rng.standard_normal((256, d // m)).astype(np.float32) for _ in range(m)]
Q = rng.standard_normal((nq, d)).astype(np.float32)
with rq.Index(C, d, devices=[0] * P) as ix:
    ix.set_codes_synth(n, 1234)
    ref = None
    for chunks in (1, 2, 4, 8):
        rq.set_tuning("IDX_QCHUNKS", chunks)
        best = 1e9
        for rep in range(4):
            dists, ids = ix.search(Q, k, id_base=0)
            best = min(best, _lib.last_timing()["kernel_ms"])
        if ref is None:
            ref = (dists.copy(), ids.copy())
        same = np.array_equal(ids, ref[1]) and np.array_equal(dists.view(np.uint32), ref[0].view(np.uint32))
        print("P=%d k=%d chunks=%d kernel_ms=%.2f same=%s" % (P, k, chunks, best, same), flush=True)
    rq.set_tuning("IDX_QCHUNKS", 0)
