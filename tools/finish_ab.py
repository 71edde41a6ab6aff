#!/usr/bin/env python3
"""Round-5 finish A/B on resident bench-shaped bases: distance buckets (SCAN_BUCKET_FINISH / SCAN_SS_MAP = 1) against select + sort /
sorted splitters (= 0), on clustered (1024 centres: mass distance ties) and bench (65536 centres) data; answers compared.
usage: python tools/finish_ab.py [K,K,...]"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
import rayuela_jl_amd as rq
import rayuela_jl_amd.synth as synth
import rayuela_jl_amd.synth_torch as st
from rayuela_jl_amd import device as rqd
from rayuela_jl_amd import _lib
dev = torch.device("cuda", 0)
def bench(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for nc in (1024, 65536):
    d, m, h, n, nq = 128, 8, 256, 1_000_000, 10_000
    gen = lambda rows, row0: st.sift_like(rows, d, seed=synth.SEED_BASE, ncentres=nc, row0=row0, device=dev)
    Q = gen(nq, 3_000_000_000); S = gen(20_000, 3_100_000_000)
    C = synth.codebooks(S.cpu().numpy(), m, h, seed=synth.SEED_CODEBOOK, iters=5, sample=20000)
    Ccat = torch.from_numpy(synth.cat_codebooks(C)).to(dev); centers = torch.from_numpy(np.stack(C)).to(dev)
    X = torch.cat([gen(250_000, o) for o in range(0, n, 250_000)], 0)
    codes = rqd.encode_pq(X, Ccat, m, h); del X
    base = rqd.order_rows(codes)
    for K in ([int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else (100, 1000, 10000)):
        out = (torch.empty((nq, K), dtype=torch.float32, device=dev), torch.empty((nq, K), dtype=torch.int32, device=dev))
        res = {}
        for mode in (1, 0):
            rq.set_tuning("SCAN_BUCKET_FINISH", mode)
            rq.set_tuning("SCAN_SS_MAP", mode)
            rq.set_tuning("SCAN_STATS", 1)
            _lib.scan_stats()
            rqd.linscan(base, centers, Q, K, out=out); torch.cuda.synchronize()
            s = _lib.scan_stats()
            rq.set_tuning("SCAN_STATS", 0)
            ms = bench(lambda: rqd.linscan(base, centers, Q, K, out=out))
            res[mode] = (out[0].clone(), out[1].clone())
            tot = sum(s[k] for k in ("lut", "sample", "stream", "final_cut", "sort_write")) or 1
            print("ncentres=%d K=%d bucket_finish=%d  %.3f ms  final_cut=%.1f%% sort_write=%.1f%% fallbacks=%d" % (nc, K, mode, ms, 100.0*s["final_cut"]/tot, 100.0*s["sort_write"]/tot, s["n_fallbacks"]), flush=True)
        print("   same:", bool(torch.equal(res[0][0].view(torch.int32), res[1][0].view(torch.int32)) and torch.equal(res[0][1], res[1][1])))
rq.set_tuning("SCAN_BUCKET_FINISH", 1)
rq.set_tuning("SCAN_SS_MAP", 1)
