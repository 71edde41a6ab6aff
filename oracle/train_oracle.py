"""numpy restatement of the training loops (TEST INFRASTRUCTURE).  Follows src/OPQ.jl:49-139 and the
Lloyd loop of src/PQ.jl:68-99 step by step with the oracle's canonical encode for the assignments;
reductions in float64 (the reference's sequential f32 loops are not pinned by any reference test)."""
import numpy as np

from . import oracle


def offsets(d, m):
    return oracle.splitarray(d, m).tolist()


def update_centers(C, RX, codes, off, h):
    """Clustering.update_centers! (call site src/OPQ.jl:121): mean of the assigned sub-vectors;
    empty clusters keep their value (see rayuela.jl_amd/train.py for the stated deviation)."""
    out = []
    for i in range(len(off) - 1):
        Xs = RX[:, off[i]:off[i + 1]].astype(np.float64)
        Ci = C[i].astype(np.float64).copy()
        for k in range(h):
            sel = codes[:, i] == k
            if sel.any():
                Ci[k] = Xs[sel].mean(0)
        out.append(Ci.astype(np.float32))
    return out


def update_centers_fast(C, RX, codes, off, h):
    """the same means through per-dimension float64 bincounts (vectorised: used where the loop above would take minutes)"""
    out = []
    for i in range(len(off) - 1):
        ci = codes[:, i].astype(np.int64)
        cnt = np.bincount(ci, minlength=h).astype(np.float64)
        Ci = C[i].astype(np.float64).copy()
        for s in range(off[i], off[i + 1]):
            sums = np.bincount(ci, weights=RX[:, s].astype(np.float64), minlength=h)
            col = s - off[i]
            Ci[cnt > 0, col] = sums[cnt > 0] / cnt[cnt > 0]
        out.append(Ci.astype(np.float32))
    return out


def reconstruct(C, codes, off, d):
    CB = np.zeros((codes.shape[0], d), dtype=np.float32)
    for i in range(len(off) - 1):
        CB[:, off[i]:off[i + 1]] = C[i][codes[:, i].astype(np.int64)]
    return CB


def train_opq(X, m, h, niter, R0, C0, fast=False):
    """R0: memory image of Julia's R (R0[i,k] = R[k,i]); C0: list of (h, sub_i).  Returns C, codes, R, obj."""
    uc = update_centers_fast if fast else update_centers
    n, d = X.shape
    off = offsets(d, m)
    R = R0.astype(np.float32)
    C = [c.astype(np.float32) for c in C0]
    cat = lambda CC: np.concatenate([c.reshape(-1) for c in CC])
    RX = oracle.rotate_T(R, X)
    codes = oracle.encode_pq(RX, cat(C), m, h)
    CB = reconstruct(C, codes, off, d)
    obj = np.zeros(niter + 1)
    for it in range(niter + 1):
        obj[it] = ((RX.astype(np.float64) - CB) ** 2).sum() / n
        G = X.astype(np.float64).T @ CB.astype(np.float64)          # X CB'
        U, _, Vt = np.linalg.svd(G, full_matrices=False)
        R = np.ascontiguousarray((U @ Vt).T.astype(np.float32))
        RX = oracle.rotate_T(R, X)
        C = uc(C, RX, codes, off, h)
        codes = oracle.encode_pq(RX, cat(C), m, h)
        CB = reconstruct(C, codes, off, d)
    return C, codes, R, obj
