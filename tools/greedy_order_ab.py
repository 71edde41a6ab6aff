#!/usr/bin/env python3
"""Round 6: the greedy balance of the row order (csrc/rq_order.hip: order_fine_greedy_kernel) against the plain 15-bit sort
(ORDER_GREEDY = 0) on the bench data: LDS passes per row and lane group (host model on the ordered codes), time of the ordering,
scan times on the prepared base and with the ordering inside the call; answers compared.   usage: python tools/greedy_order_ab.py [8|16] [n] [nq]"""
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
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000


def bench(fn, iters=10, warm=3):
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


def group_passes(g):
    G, _, m = g.shape
    out = np.zeros((G, m))
    for k in range(m):
        pres = np.zeros((G, 256), bool)
        pres[np.arange(G)[:, None], g[:, :, k].astype(np.int64)] = True
        out[:, k] = pres.reshape(G, 8, 32).sum(1).max(1)
    return out


rpt, blk = {4: (4, 16384), 8: (2, 8192), 16: (1, 4096)}[M]
if M == 16:
    d, h = 96, 256
    gen = lambda rows, row0: st.deep_like(rows, d, seed=synth.SEED_BASE, row0=row0, device=dev)                   # noqa: E731
    R = torch.from_numpy(synth.rotation(d)).to(dev)
else:
    d, h = 128, 256
    gen = lambda rows, row0: st.sift_like(rows, d, seed=synth.SEED_BASE, ncentres=65536, row0=row0, device=dev)   # noqa: E731
    R = None
Q = gen(nq, 3_000_000_000)
S = gen(20_000, 3_100_000_000)
if R is not None:
    Q, S = rqd.rotate_T(R, Q), rqd.rotate_T(R, S)
C = synth.codebooks(S.cpu().numpy(), M, h, seed=synth.SEED_CODEBOOK, iters=5, sample=20000)
Ccat = torch.from_numpy(synth.cat_codebooks(C)).to(dev)
centers = torch.from_numpy(np.stack(C)).to(dev)
X = torch.cat([gen(min(250_000, n - o), o) for o in range(0, n, 250_000)], 0)
if R is not None:
    X = rqd.rotate_T(R, X)
codes = rqd.encode_pq(X, Ccat, M, h)
del X

tile, stride = 64 * rpt, 16
pos = np.arange(n, dtype=np.int64)
blkid = pos // blk
is_sample = (blkid % stride == 0) & (blkid // stride < n // (stride * blk))
u = pos % tile
gid = (pos // tile) * (2 * rpt) + ((u // rpt) // 32) * rpt + (u % rpt)
movable = (~is_sample) & (pos < (n // tile - 2) * tile)
sel = np.flatnonzero(movable)
osel = sel[np.argsort(gid[sel], kind="stable")]
bases, t_order = {}, {}
for g in (0, 1):
    rq.set_tuning("ORDER_GREEDY", g)
    bases[g] = rqd.order_rows(codes)
    t_order[g] = bench(lambda: rqd.order_rows(codes), 10)
    c = bases[g].codes.cpu().numpy()
    pm = bases[g].perm.cpu().numpy()
    assert np.array_equal(np.sort(pm), np.arange(n)), "not a permutation"
    assert np.array_equal(c[:, :M], codes.cpu().numpy()[pm][:, :M]), "codes do not follow perm"
    p = group_passes(c[osel][: len(osel) // 32 * 32].reshape(-1, 32, c.shape[1])[:, :, :M])
    print("m=%d n=%d ORDER_GREEDY=%d: order_rows %.4f ms; passes per table %s  sum %.2f" % (M, n, g, t_order[g], p.mean(0).round(2), p.sum(1).mean()), flush=True)
for K in (1, 100, 1000):
    out = (torch.empty((nq, K), dtype=torch.float32, device=dev), torch.empty((nq, K), dtype=torch.int32, device=dev))
    res, outs = {}, {}
    for rep in range(2):
        for g in (0, 1):
            res.setdefault(("prep", g), []).append(bench(lambda: rqd.linscan(bases[g], centers, Q, K, out=out)))
            outs[g] = (out[0].clone(), out[1].clone())
            rq.set_tuning("ORDER_GREEDY", g)
            res.setdefault(("call", g), []).append(bench(lambda: rqd.linscan(codes, centers, Q, K, out=out)))
    same = bool(torch.equal(outs[0][0].view(torch.int32), outs[1][0].view(torch.int32)) and torch.equal(outs[0][1], outs[1][1]))
    mn = {k: min(v) for k, v in res.items()}
    print("m=%d K=%-5d prepared: plain %.4f  greedy %.4f ms (%.1f %%)   in-call: plain %.4f  greedy %.4f ms (%.1f %%)   same answer: %s" % (
        M, K, mn[("prep", 0)], mn[("prep", 1)], 100 * (mn[("prep", 1)] / mn[("prep", 0)] - 1), mn[("call", 0)], mn[("call", 1)],
        100 * (mn[("call", 1)] / mn[("call", 0)] - 1), same), flush=True)
rq.set_tuning("ORDER_GREEDY", 1)
