#!/usr/bin/env python3
"""PROBE (round 6): how much would the scan gain if the rows of a sort bucket were BALANCED over the bucket's lane groups by a
greedy pass over the tables the sort key does not cover?  The base is ordered by the library with a key that is 3 bits shorter
(ORDER_BITS), then the rows of every bucket are re-dealt on the HOST (numpy, the algorithm a GPU kernel would run: arrival order,
cost = sum over the free tables of the marginal increase of (column load)^2, equal values are free) among the slots the bucket
holds, and the scan is timed on both bases.  Answers must be identical.   usage: python tools/greedy_order_probe.py [8|16]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import rayuela_jl_amd as rq                      # noqa: E402
import rayuela_jl_amd.synth as synth             # noqa: E402
import rayuela_jl_amd.synth_torch as st          # noqa: E402
from rayuela_jl_amd import device as rqd, _lib   # noqa: E402

dev = torch.device("cuda", 0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 4          # tables under the sort key (3 bits each)
n, nq = 1_000_000, 10_000


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


if M == 8:
    d, h, rpt, blk = 128, 256, 2, 8192
    gen = lambda rows, row0: st.sift_like(rows, d, seed=synth.SEED_BASE, ncentres=65536, row0=row0, device=dev)   # noqa: E731
    R = None
else:
    d, h, rpt, blk = 96, 256, 1, 4096
    gen = lambda rows, row0: st.deep_like(rows, d, seed=synth.SEED_BASE, row0=row0, device=dev)                   # noqa: E731
    R = torch.from_numpy(synth.rotation(d)).to(dev)
Q = gen(nq, 3_000_000_000)
S = gen(20_000, 3_100_000_000)
if R is not None:
    Q, S = rqd.rotate_T(R, Q), rqd.rotate_T(R, S)
C = synth.codebooks(S.cpu().numpy(), M, h, seed=synth.SEED_CODEBOOK, iters=5, sample=20000)
Ccat = torch.from_numpy(synth.cat_codebooks(C)).to(dev)
centers = torch.from_numpy(np.stack(C)).to(dev)
X = torch.cat([gen(250_000, o) for o in range(0, n, 250_000)], 0)
if R is not None:
    X = rqd.rotate_T(R, X)
codes = rqd.encode_pq(X, Ccat, M, h)
del X

base15 = rqd.order_rows(codes)                       # the shipped order (15 key bits at 1e6 rows)
rq.set_tuning("ORDER_GREEDY", 0)
base15 = rqd.order_rows(codes)                       # (plain sort: the comparison base)
rq.set_tuning("ORDER_BITS", 3 * NT)
base12 = rqd.order_rows(codes)                       # 4 tables x 3 bits: the buckets the greedy pass works in
rq.set_tuning("ORDER_BITS", 0)
rq.set_tuning("ORDER_GREEDY", 1)
baseG = rqd.order_rows(codes)                        # the shipped order (12 bits + device balance)
torch.cuda.synchronize()

tile, stride = 64 * rpt, 16
oc = base12.codes.cpu().numpy().copy()
pm = base12.perm.cpu().numpy().copy()
pos = np.arange(n, dtype=np.int64)
blkid = pos // blk
sgroups = n // (stride * blk)
is_sample = (blkid % stride == 0) & (blkid // stride < sgroups)
# lane group of a position: (tile, b, r) with u = pos % tile, lane = u // rpt, b = lane // 32, r = u % rpt
u = pos % tile
gid = (pos // tile) * (2 * rpt) + ((u // rpt) // 32) * rpt + (u % rpt)
nsorted = n - int(is_sample.sum())
movable = ~is_sample
# (the ragged last tile of the sorted index space stays in sort order: leave the last 2 tiles of positions alone)
movable &= pos < (n // tile - 2) * tile
key = np.zeros(n, np.int64)
for k in range(NT):
    key = (key << 3) | (oc[:, k].astype(np.int64) >> 5)
free = list(range(NT, M))
t0 = time.time()
idx = np.flatnonzero(movable)
order = np.lexsort((pm[idx], key[idx]))              # bucket-major, arrival order inside a bucket
idx = idx[order]
kk = key[idx]
bnd = np.flatnonzero(np.r_[True, kk[1:] != kk[:-1], True])
new_oc, new_pm = oc.copy(), pm.copy()
T = len(free)
for bi in range(len(bnd) - 1):
    p_b = idx[bnd[bi]:bnd[bi + 1]]                   # positions this bucket holds
    rows_c, rows_p = oc[p_b], pm[p_b]
    g_b = gid[p_b]
    ug, inv, cap = np.unique(g_b, return_inverse=True, return_counts=True)
    G = len(ug)
    slots = [list(p_b[inv == g]) for g in range(G)]
    col = (rows_c[:, free] & 31).astype(np.int64)
    val = rows_c[:, free].astype(np.int64)
    load = np.zeros((G, T, 32), np.int32)
    seen = np.zeros((G, T, 256), bool)
    fill = np.zeros(G, np.int32)
    tr = np.arange(T)
    for r in range(len(p_b)):
        add = ((~seen[:, tr, val[r]]) * (2 * load[:, tr, col[r]] + 1)).sum(1).astype(np.float64)
        add[fill >= cap] = 1e9
        g = int(np.argmin(add))
        dst = slots[g][fill[g]]
        fill[g] += 1
        new_oc[dst], new_pm[dst] = rows_c[r], rows_p[r]
        fresh = ~seen[g, tr, val[r]]
        seen[g, tr, val[r]] = True
        load[g, tr[fresh], col[r][fresh]] += 1
print("greedy re-deal of %d rows in %d buckets on the host: %.1f s" % (len(idx), len(bnd) - 1, time.time() - t0), flush=True)
assert np.array_equal(np.sort(new_pm), np.sort(pm))


def passes_of(c):
    sel = np.flatnonzero(movable)
    o = sel[np.argsort(gid[sel], kind="stable")]
    g = c[o][: len(o) // 32 * 32].reshape(-1, 32, c.shape[1])[:, :, :M]
    return group_passes(g)


for name, c in (("plain 15-bit sort", base15.codes.cpu().numpy()), ("shipped (device balance)", baseG.codes.cpu().numpy()), ("%d-bit order" % (3 * NT), oc), ("%d-bit order + host greedy" % (3 * NT), new_oc)):
    p = passes_of(c)
    print("%-24s passes per table %s  sum %.2f" % (name, p.mean(0).round(2), p.sum(1).mean()), flush=True)
base12.codes.copy_(torch.from_numpy(new_oc).to(dev))
base12.perm.copy_(torch.from_numpy(new_pm).to(dev))
torch.cuda.synchronize()
for K in (1, 100, 1000):
    out = (torch.empty((nq, K), dtype=torch.float32, device=dev), torch.empty((nq, K), dtype=torch.int32, device=dev))
    res = {}
    for rep in range(2):
        for name, b in (("shipped", baseG), ("greedy", base12)):
            ms = bench(lambda: rqd.linscan(b, centers, Q, K, out=out))
            res.setdefault(name, []).append(ms)
            res[name + "_out"] = (out[0].clone(), out[1].clone())
    same = bool(torch.equal(res["shipped_out"][0].view(torch.int32), res["greedy_out"][0].view(torch.int32)) and torch.equal(res["shipped_out"][1], res["greedy_out"][1]))
    rq.set_tuning("SCAN_STATS", 1)
    fb = {}
    for name, b in (("shipped", baseG), ("greedy", base12)):
        _lib.scan_stats()
        rqd.linscan(b, centers, Q, K, out=out)
        torch.cuda.synchronize()
        fb[name] = _lib.scan_stats()["n_fallbacks"]
    rq.set_tuning("SCAN_STATS", 0)
    print("m=%d K=%-5d shipped (12 bits + device balance) %.4f ms   host greedy %.4f ms   (%.1f %%)  same answer: %s  fallbacks %s" % (
        M, K, min(res["shipped"]), min(res["greedy"]), 100.0 * (min(res["greedy"]) / min(res["shipped"]) - 1.0), same, fb), flush=True)
