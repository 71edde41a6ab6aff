#!/usr/bin/env python3
"""The last partial round of work items: 1e4 queries = 1250 groups over 512 resident workgroups leave 226 whole-base items for a third
round that fills less than half of the slots.  SCAN_TAIL_SLICES cuts the groups beyond a multiple of the CU count into row slices
(merged by merge_topk).  Same-box A/B on the bench data, prepared base and in-call ordering.   usage: python tools/tail_slices_ab.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import rayuela_jl_amd as rq                      # noqa: E402
import rayuela_jl_amd.synth as synth             # noqa: E402
import rayuela_jl_amd.synth_torch as st          # noqa: E402
from rayuela_jl_amd import device as rqd, _lib   # noqa: E402

dev = torch.device("cuda", 0)


def bench(fn, iters=20, warm=3):
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


d, m, h, n = 128, 8, 256, 1_000_000
gen = lambda rows, row0: st.sift_like(rows, d, seed=synth.SEED_BASE, ncentres=65536, row0=row0, device=dev)   # noqa: E731
S = gen(20_000, 3_100_000_000)
C = synth.codebooks(S.cpu().numpy(), m, h, seed=synth.SEED_CODEBOOK, iters=5, sample=20000)
Ccat = torch.from_numpy(synth.cat_codebooks(C)).to(dev)
centers = torch.from_numpy(np.stack(C)).to(dev)
X = torch.cat([gen(250_000, o) for o in range(0, n, 250_000)], 0)
codes = rqd.encode_pq(X, Ccat, m, h)
del X
prepared = rqd.order_rows(codes)
for nq in (10_000, 7_000, 5_000, 3_000):
    Q = gen(nq, 3_000_000_000)
    for K in (1, 100, 1000):
        out = (torch.empty((nq, K), dtype=torch.float32, device=dev), torch.empty((nq, K), dtype=torch.int32, device=dev))
        ref = None
        line = []
        for ts in (0, 2, 3, 4, 0):
            rq.set_tuning("SCAN_TAIL_SLICES", ts)
            ms = bench(lambda: rqd.linscan(prepared, centers, Q, K, out=out))
            if ref is None:
                ref = (out[0].clone(), out[1].clone())
            same = bool(torch.equal(out[0].view(torch.int32), ref[0].view(torch.int32)) and torch.equal(out[1], ref[1]))
            pl = _lib.scan_plan(n, nq, m, d, K)
            line.append("ts=%d: %.4f ms%s (whole %d, slices %d)" % (ts, ms, "" if same else " MISMATCH", pl["whole"], pl["slices"]))
        rq.set_tuning("SCAN_TAIL_SLICES", 0)
        print("nq=%d K=%d  " % (nq, K) + " | ".join(line), flush=True)
