#!/usr/bin/env python3
"""Same-box A/B of two library builds on the scan shapes that matter: SIFT1M-shape k = 1 / 1000 (prepared base) and the 1.25e8-row
shard of config 5 (1024 queries, k = 100).  Each library runs in its own process, alternating A B A B.
usage: python tools/ab_shard.py libA.so libB.so [libC.so ...] [rounds]"""
import json
import os
import subprocess
import sys

CHILD = r'''
import os, sys, json, numpy as np, torch
sys.path.insert(0, os.getcwd())
import rayuela_jl_amd as rq
import rayuela_jl_amd.synth as synth
import rayuela_jl_amd.synth_torch as st
from rayuela_jl_amd import device as rqd
dev = torch.device("cuda", 0)
def bench(fn, iters, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
d, m, h = 128, 8, 256
gen = lambda rows, row0: st.sift_like(rows, d, seed=synth.SEED_BASE, ncentres=65536, row0=row0, device=dev)
S = gen(20_000, 3_100_000_000)
C = synth.codebooks(S.cpu().numpy(), m, h, seed=synth.SEED_CODEBOOK, iters=5, sample=20000)
Ccat = torch.from_numpy(synth.cat_codebooks(C)).to(dev); centers = torch.from_numpy(np.stack(C)).to(dev)
out = {}
n, nq = 1_000_000, 10_000
Q = gen(nq, 3_000_000_000)
X = torch.cat([gen(250_000, o) for o in range(0, n, 250_000)], 0)
codes = rqd.encode_pq(X, Ccat, m, h); del X
base = rqd.order_rows(codes)
for K in (1, 1000):
    o = (torch.empty((nq, K), dtype=torch.float32, device=dev), torch.empty((nq, K), dtype=torch.int32, device=dev))
    out["sift1m_k%d_prepared" % K] = bench(lambda: rqd.linscan(base, centers, Q, K, out=o), 10)
o = (torch.empty((nq, 10000), dtype=torch.float32, device=dev), torch.empty((nq, 10000), dtype=torch.int32, device=dev))
out["sift1m_k10000_arrival"] = bench(lambda: rqd.linscan(codes, centers, Q, 10000, out=o), 5)
del o, base, codes
ns = 125_000_000
big = rqd.order_rows(rqd.synth_codes(ns, m, synth.SEED_BASE, row0=0, device=dev))
Q2 = Q[:1024].contiguous()
o = (torch.empty((1024, 100), dtype=torch.float32, device=dev), torch.empty((1024, 100), dtype=torch.int32, device=dev))
out["shard_1.25e8_k100"] = bench(lambda: rqd.linscan(big, centers, Q2, 100, out=o), 4)
print("RESULT " + json.dumps(out))
'''
libs = [a for a in sys.argv[1:] if a.endswith('.so')]
rounds = int(sys.argv[-1]) if not sys.argv[-1].endswith('.so') else 2
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, RAYUELA_HIP_LIB=os.path.abspath(l), RAYUELA_HIP_LENIENT="1")
        p = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        line = [x for x in p.stdout.splitlines() if x.startswith("RESULT ")]
        if not line:
            print("FAILED", l, p.stderr[-400:])
            continue
        res[l].append(json.loads(line[0][7:]))
        print(os.path.basename(l), json.dumps({k: round(v, 4) for k, v in res[l][-1].items()}), flush=True)
print()
for k in res[libs[0]][0]:
    print("%-26s " % k + "   ".join("%s: min %.4f ms" % (os.path.basename(l), min(x[k] for x in res[l])) for l in libs))
