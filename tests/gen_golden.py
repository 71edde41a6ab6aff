"""Regenerates tests/golden/*.npz.  Run in the BUILD container only (needs /root/reference):

    make -C oracle all ref && python tests/gen_golden.py

Scan fixtures: inputs + the outputs of the REAL reference scan, i.e. oracle/_ref/linscan_aqd.so
compiled from /root/reference/deps/src/linscan_aqd.cpp with the reference flags
(deps/build.jl:23).  These pin oracle/rq_oracle.c and the HIP path (ids and distances bit-exact).

Encode fixtures: the reference encode is Julia + Distances.jl + Clustering.jl + OpenBLAS and
cannot run here (no Julia, packages not vendored) -> PARITY UNPINNED.  What is stored is the
canonical f32 result (oracle/rq_oracle.c, fmaf-chain order), an independent float64 evaluation of
the same formula (codes64) and the f64 top-2 gap per item so a mismatch can be classified as a
near-tie.  recall fixture: computed by hand-rolled loops that follow src/Linscan.jl:206-230.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
import rayuela_jl_amd.synth as synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def scan_case(name, n, m, sub, nq, Ks, kind):
    d = m * sub
    if kind == "sift":
        Xb = synth.sift_like(max(n, 4096), d, seed=11)
        Xq = synth.sift_like(nq, d, seed=12)
        C = synth.codebooks(Xb, m, 256, seed=13, iters=2, sample=4096)
        centers = np.stack(C)  # [m][256][sub]
        codes = oracle.encode_pq(Xb[:n], synth.cat_codebooks(C), m, 256)
    elif kind == "deep":
        Xb = synth.deep_like(max(n, 4096), d, seed=21)
        Xq = synth.deep_like(nq, d, seed=22)
        C = synth.codebooks(Xb, m, 256, seed=23, iters=2, sample=4096)
        centers = np.stack(C)
        codes = oracle.encode_pq(Xb[:n], synth.cat_codebooks(C), m, 256)
    elif kind == "ties":  # all rows identical -> ids 0..K-1
        centers = (synth.splitmix64(np.arange(m * 256 * sub, dtype=np.uint64) ^ np.uint64(5))
                   % np.uint64(7)).astype(np.float32).reshape(m, 256, sub)
        Xq = (synth.splitmix64(np.arange(nq * d, dtype=np.uint64) ^ np.uint64(6))
              % np.uint64(5)).astype(np.float32).reshape(nq, d)
        codes = np.tile(synth.random_codes(1, m, seed=3), (n, 1))
    elif kind == "dups":  # integer LUTs, few distinct codes -> massive distance ties
        centers = (synth.splitmix64(np.arange(m * 256 * sub, dtype=np.uint64) ^ np.uint64(8))
                   % np.uint64(3)).astype(np.float32).reshape(m, 256, sub)
        Xq = (synth.splitmix64(np.arange(nq * d, dtype=np.uint64) ^ np.uint64(9))
              % np.uint64(3)).astype(np.float32).reshape(nq, d)
        codes = (synth.random_codes(n, m, seed=4) % 4).astype(np.uint8)
    else:
        raise ValueError(kind)
    data = dict(codes=codes, centers=centers.astype(np.float32), queries=Xq.astype(np.float32),
                Ks=np.asarray(Ks, dtype=np.int32))
    for K in Ks:
        dists, ids = oracle.ref_linscan_aqd_query(codes, centers, Xq, K)
        data["dists_K%d" % K] = dists
        data["ids_K%d" % K] = ids
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
    print(name, codes.shape, Ks)


def aq_case(name, n, m, d, nq, Ks, seed):
    """LSQ / CQ scans: outputs of the REAL reference linscan_aqd_pairwise_byte.so."""
    rng = np.random.default_rng(seed)
    codebooks = (rng.standard_normal((m * 256, d)) * 3).astype(np.float32)
    queries = (rng.standard_normal((nq, d)) * 3).astype(np.float32)
    codes = synth.random_codes(n, m, seed=seed)
    codes[n // 2:n // 2 + 40] = codes[7]                     # duplicate rows -> exact distance ties
    recon = sum(codebooks[k * 256 + codes[:, k].astype(np.int64)] for k in range(m))
    dbnorms = (recon.astype(np.float64) ** 2).sum(1).astype(np.float32)
    data = dict(codes=codes, codebooks=codebooks, queries=queries, dbnorms=dbnorms, Ks=np.asarray(Ks, np.int32))
    for K in Ks:
        dl, il = oracle.linscan_lsq(codes, codebooks, queries, dbnorms, K, use_ref=True)
        dc, ic = oracle.linscan_cq(codes, codebooks, queries, K, use_ref=True)
        data.update({"lsq_d%d" % K: dl, "lsq_i%d" % K: il, "cq_d%d" % K: dc, "cq_i%d" % K: ic})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
    print(name, codes.shape, Ks)


def encode64(X, C, m):
    """float64 evaluation of the same formula (independent of the f32 order)."""
    n, d = X.shape
    off = synth.splitarray(d, m)
    codes = np.zeros((n, m), dtype=np.uint8)
    gap = np.zeros((n, m), dtype=np.float64)
    for i in range(m):
        Xs = X[:, off[i]:off[i + 1]].astype(np.float64)
        Ci = C[i].astype(np.float64)
        v = (Ci * Ci).sum(1)[None, :] + (Xs * Xs).sum(1)[:, None] - 2.0 * Xs @ Ci.T
        v = np.maximum(v, 0.0)
        codes[:, i] = v.argmin(1)
        part = np.partition(v, 1, axis=1)
        gap[:, i] = part[:, 1] - part[:, 0]
    return codes, gap


def encode_case(name, n, d, m, h, kind, with_R):
    if kind == "sift":
        X = synth.sift_like(n + 2048, d, seed=31)
    else:
        X = synth.deep_like(n + 2048, d, seed=32)
    C = synth.codebooks(X, m, h, seed=33, iters=2, sample=2048)
    X = X[:n].copy()
    if kind == "sift":
        # exact hits: some vectors ARE a centroid in some subspace -> v = 0 after the clamp
        off = synth.splitarray(d, m)
        for j in range(0, min(n, 64)):
            i = j % m
            X[j, off[i]:off[i + 1]] = C[i][(j * 7) % h]
    data = dict(X=X, C=synth.cat_codebooks(C), m=np.int32(m), h=np.int32(h))
    if with_R:
        R = synth.rotation(d, seed=34)
        data["R"] = R
        data["codes_opq"] = oracle.encode_opq(X, R, data["C"], m, h)
        # RX itself is not stored (size); tests recompute it with the oracle and check it vs f64
    codes, costs = oracle.encode_pq(X, data["C"], m, h, with_costs=True)
    c64, gap = encode64(X, C, m)
    data.update(codes=codes, costs=costs, codes64=c64, gap64=gap.astype(np.float32))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
    print(name, X.shape, "f32-vs-f64 code flips:", int((codes != c64).sum()))


def rvq_case(name, n, d, m, h, kind):
    """quantize_rvq (src/RVQ.jl:18-66): canonical f32 codes/counts/residual of the oracle (PARITY UNPINNED
    like the PQ encode) + a float64 replay (codes64 follow the f32 path's residual; gap64 = f64 top-2 gap)."""
    X = synth.sift_like(n + 2048, d, seed=51) if kind == "sift" else synth.deep_like(n + 2048, d, seed=52)
    C = synth.rvq_codebooks(X, m, h, seed=53)
    X = X[:n].copy()
    X[3] = C[0][5]                       # exact hit in stage 1: distance clamps to 0
    codes, counts, Xr = oracle.encode_rvq(X, C, with_extras=True)
    R = X.copy()
    c64 = np.zeros_like(codes)
    gap = np.zeros(codes.shape, dtype=np.float32)
    for i in range(m):
        d2 = ((R[:, None, :].astype(np.float64) - C[i][None].astype(np.float64)) ** 2).sum(-1)
        c64[:, i] = d2.argmin(1)
        part = np.partition(d2, 1, axis=1)
        gap[:, i] = (part[:, 1] - part[:, 0]).astype(np.float32)
        R = R - C[i][codes[:, i]]
    assert np.array_equal(R.view(np.uint32), Xr.view(np.uint32))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), X=X, C=C, codes=codes, counts=counts, Xr=Xr,
                        codes64=c64, gap64=gap)
    print(name, X.shape, "f32-vs-f64 code flips:", int((codes != c64).sum()), "unused:", int((counts == 0).sum()))


def recall_case():
    """src/Linscan.jl:196-234 evaluated with explicit loops (1-based ids like the Julia caller)."""
    k, nq = 10, 6
    gt = np.array([5, 9, 3, 100, 7, 42], dtype=np.uint32)
    idx = np.zeros((nq, k), dtype=np.uint32)
    idx[:] = np.arange(200, 200 + k, dtype=np.uint32)[None, :]
    idx[0, 0] = 5          # rank 1
    idx[1, 4] = 9          # rank 5
    idx[2, 9] = 3          # rank 10
    #  query 3: absent -> k+1
    idx[4, 1] = 7; idx[4, 6] = 7   # occurs twice -> k+1 (length(nn_pos) != 1)
    idx[5, 2] = 42         # rank 3
    ranks = []
    for i in range(nq):
        pos = [j + 1 for j in range(k) if idx[i, j] == gt[i]]
        ranks.append(pos[0] if len(pos) == 1 else k + 1)
    recall = np.zeros(k)
    for R in range(1, k + 1):
        recall[R - 1] = sum(1 for r in ranks if r <= R and r <= k) / nq
    np.savez_compressed(os.path.join(OUT, "recall_tiny.npz"), gt=gt, idx=idx, k=np.int32(k),
                        recall=recall, ranks=np.asarray(ranks))
    print("recall_tiny", recall)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) < 2 or sys.argv[1] == "rvq":
        rvq_case("rvq_sift_mini", 600, 128, 4, 256, "sift")
        rvq_case("rvq_deep_mini", 500, 96, 3, 100, "deep")
        if len(sys.argv) >= 2:
            sys.exit(0)
    assert oracle.ref_available(), "build oracle/_ref first: make -C oracle ref"
    scan_case("scan_sift_mini", 4096, 8, 16, 16, [1, 10, 100, 1000], "sift")
    scan_case("scan_deep_mini", 4096, 16, 6, 8, [1, 10, 100], "deep")
    scan_case("scan_all_ties", 3000, 8, 4, 4, [1, 17, 1000], "ties")
    scan_case("scan_dups", 5000, 4, 2, 5, [1, 64, 2500], "dups")
    scan_case("scan_k_eq_n", 777, 8, 2, 3, [777], "dups")
    encode_case("encode_sift_mini", 1024, 128, 8, 256, "sift", with_R=True)
    encode_case("encode_deep_mini", 1024, 96, 16, 256, "deep", with_R=True)
    encode_case("encode_uneven", 512, 10, 4, 64, "sift", with_R=False)   # splitarray -> 3,3,2,2
    encode_case("encode_h100", 512, 32, 4, 100, "deep", with_R=False)    # h not a multiple of 32
    recall_case()
    assert oracle.ref_aq_available(), "make -C oracle ref"
    aq_case("aq_mini_m8", 3000, 8, 32, 6, [1, 100, 1000], seed=41)
    aq_case("aq_mini_m4", 1500, 4, 16, 5, [10, 1500], seed=42)
