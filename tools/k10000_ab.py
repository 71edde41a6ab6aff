#!/usr/bin/env python3
"""k = 10000 (the reference's default, src/Linscan.jl:10) on the bench data: same-box A/B of the scan's knobs -- FINE vs coarse byte
tables, arrival order vs in-call ordering vs a prepared base -- for the library named by RAYUELA_HIP_LIB (build variants:
tools/build_variant.sh).  Every answer is compared with the first one.   usage: python tools/k10000_ab.py [K,K,...] [iters]"""
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
Ks = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [10000]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10


def bench(fn, warm=3):
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


d, m, h, n, nq = 128, 8, 256, 1_000_000, 10_000
gen = lambda rows, row0: st.sift_like(rows, d, seed=synth.SEED_BASE, ncentres=65536, row0=row0, device=dev)   # noqa: E731
Q = gen(nq, 3_000_000_000)
S = gen(20_000, 3_100_000_000)
C = synth.codebooks(S.cpu().numpy(), m, h, seed=synth.SEED_CODEBOOK, iters=5, sample=20000)
Ccat = torch.from_numpy(synth.cat_codebooks(C)).to(dev)
centers = torch.from_numpy(np.stack(C)).to(dev)
X = torch.cat([gen(250_000, o) for o in range(0, n, 250_000)], 0)
codes = rqd.encode_pq(X, Ccat, m, h)
del X
prepared = rqd.order_rows(codes)
print("lib", (_lib.lib().rq_version() or b"").decode(), os.environ.get("RAYUELA_HIP_LIB", "(in-tree)"))
for K in Ks:
    out = (torch.empty((nq, K), dtype=torch.float32, device=dev), torch.empty((nq, K), dtype=torch.int32, device=dev))
    ref = None
    for name, base, tun in (("arrival, default tables", codes, {"SCAN_ORDER": 0}),
                            ("arrival, coarse tables", codes, {"SCAN_ORDER": 0, "SCAN_FINE_MIN_K": 1 << 20}),
                            ("arrival, fine tables", codes, {"SCAN_ORDER": 0, "SCAN_FINE_MIN_K": 1}),
                            ("ordered in call", codes, {"SCAN_ORDER": 2}),
                            ("prepared base", prepared, {}),
                            ("arrival, default tables", codes, {"SCAN_ORDER": 0})):
        for k, v in tun.items():
            rq.set_tuning(k, v)
        rq.set_tuning("SCAN_STATS", 1)
        _lib.scan_stats()
        rqd.linscan(base, centers, Q, K, out=out)
        torch.cuda.synchronize()
        s = _lib.scan_stats()
        rq.set_tuning("SCAN_STATS", 0)
        ms = bench(lambda: rqd.linscan(base, centers, Q, K, out=out))
        kern = (_lib.lib().rq_last_scan_kernel() or b"").decode()
        same = True
        if ref is None:
            ref = (out[0].clone(), out[1].clone())
        else:
            same = bool(torch.equal(out[0].view(torch.int32), ref[0].view(torch.int32)) and torch.equal(out[1], ref[1]))
        tot = sum(s[k] for k in ("lut", "sample", "stream", "final_cut", "sort_write")) or 1
        print("K=%-6d %-26s %7.3f ms  %-40s stream=%.1f%% finish=%.1f%% alive(first block)=%.2f%% fallbacks=%d same=%s" % (
            K, name, ms, kern, 100.0 * s["stream"] / tot, 100.0 * s["sort_write"] / tot,
            100.0 * s["first_block_pushed"] / max(1, s["first_block_rows"]), s["n_fallbacks"], same), flush=True)
        for k in tun:
            rq.set_tuning(k, {"SCAN_ORDER": 1, "SCAN_FINE_MIN_K": 0}[k])
