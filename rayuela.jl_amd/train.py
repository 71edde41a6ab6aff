"""Device-resident variant of the training loops (torch tensors in, nothing leaves the GPU between steps).
The drop-in entry points are rq_train_pq / rq_train_opq (csrc/rq_train_host.hip) behind PQ.train_pq and
OPQ.train_opq; this module runs the same loop from Python over the rq_dev_* pieces and is kept as the
readable reference of the step order: train_pq (src/PQ.jl:68-99) and train_opq (src/OPQ.jl:49-139).

Host code drives; every O(n) step runs on the device through the C ABI: assignments = rq_dev_encode_pq
(the hot-path kernel), R'X = rq_dev_rotate_T, update_centers / reconstruct / qerror / gram = rq_train.hip.
The d x d SVD (src/OPQ.jl:112-113) stays on the host like in the reference (LAPACK via numpy).

Deviations from the reference, both forced by its use of Julia's global RNG and stated here:
  * initial centres: `sample(1:n, h, replace=false)` (OPQ.jl:81-84) / k-means++ (PQ.jl:86, Clustering.jl)
    are replaced by h distinct rows drawn with a seeded numpy Generator (same distribution as OPQ's);
  * an empty cluster keeps its previous centre (Clustering.update_centers! would multiply by 1/0);
    train_pq additionally re-seeds empty clusters from sampled points, as Clustering.kmeans repicks them.
Returned arrays follow the memory-image convention of the other mirrors: C list of (h, sub_i), B (n, m)
int16 one-based, R (d, d) with R[i, k] == R_julia[k, i].
"""
import numpy as np
import torch

from . import device as rqd
from .utils import _as_f32, cat_codebooks


def _split(Ccat, off, h):
    out, pos = [], 0
    C = Ccat.detach().cpu().numpy()
    for i in range(len(off) - 1):
        sub = int(off[i + 1] - off[i])
        out.append(C[pos:pos + h * sub].reshape(h, sub).copy())
        pos += h * sub
    return out


def _offsets(d, m):
    per, extra = divmod(d, m)
    off = [0]
    for i in range(m):
        off.append(off[-1] + per + (1 if i < extra else 0))
    return off


def _init_centers(RX, off, h, rng):
    n = RX.shape[0]
    parts = []
    for i in range(len(off) - 1):
        perm = torch.from_numpy(rng.choice(n, size=h, replace=False)).to(RX.device)
        parts.append(RX[perm][:, off[i]:off[i + 1]].contiguous().reshape(-1))
    return torch.cat(parts).contiguous()


def train_pq(X, m, h, niter=25, V=False, seed=0, device="cuda"):
    """train_pq(X, m, h, niter=25, V=false) -> C, B, error        (src/PQ.jl:68-99)

    Lloyd's k-means in every subspace, all m subspaces per device pass."""
    X = _as_f32(X, "X")
    n, d = X.shape
    off = _offsets(d, m)
    rng = np.random.default_rng(seed)
    Xd = torch.from_numpy(X).to(device)
    Ccat = _init_centers(Xd, off, h, rng)
    codes = torch.empty((n, m), dtype=torch.uint8, device=device)
    prev = None
    for it in range(niter):
        rqd.encode_pq(Xd, Ccat, m, h, out=codes)                 # update_assignments!
        if prev is not None and torch.equal(prev, codes):
            if V:
                print("  converged after %d iterations" % it)
            break
        prev = codes.clone()
        counts = rqd.update_centers(Ccat, Xd, codes, m, h)        # update_centers!
        empty = (counts == 0).nonzero().cpu().numpy()
        for (i, k) in empty:                                      # repick unused centres
            sub = off[i + 1] - off[i]
            row = int(rng.integers(n))
            base = h * off[i] + int(k) * sub
            Ccat[base:base + sub] = Xd[row, off[i]:off[i + 1]]
        if V:
            print("  iter %d: %d empty clusters" % (it, len(empty)))
    rqd.encode_pq(Xd, Ccat, m, h, out=codes)
    err = rqd.qerror(Xd, rqd.reconstruct(codes, Ccat, d, h))     # qerror_pq, src/qerrors.jl:93-100
    B = (codes.cpu().numpy().astype(np.int16) + 1)
    return _split(Ccat, off, h), B, err


def train_opq(X, m, h, niter, init, V=False, seed=0, device="cuda", R0=None, C0=None):
    """train_opq(X, m, h, niter, init, V=false) -> C, B, R, obj   (src/OPQ.jl:49-139)

    R0 / C0 (optional) override the random initialisation so a run can be reproduced exactly
    (tests pass the oracle's init)."""
    X = _as_f32(X, "X")
    n, d = X.shape
    off = _offsets(d, m)
    rng = np.random.default_rng(seed)
    if R0 is not None:
        R = _as_f32(R0, "R0").copy()
    elif init == "natural":
        R = np.eye(d, dtype=np.float32)
    elif init == "random":
        U, _, _ = np.linalg.svd(rng.standard_normal((d, d)).astype(np.float32))
        R = np.ascontiguousarray(U.T.astype(np.float32))         # memory image of Julia's R = U
    else:
        raise ValueError("Intialization %s unknown" % init)
    Xd = torch.from_numpy(X).to(device)
    Rd = torch.from_numpy(R).to(device)
    RX = rqd.rotate_T(Rd, Xd)
    Ccat = torch.from_numpy(cat_codebooks(C0)).to(device) if C0 is not None else _init_centers(RX, off, h, rng)
    codes = torch.empty((n, m), dtype=torch.uint8, device=device)
    rqd.encode_pq(RX, Ccat, m, h, out=codes)
    CB = rqd.reconstruct(codes, Ccat, d, h)
    obj = np.zeros(niter + 1, dtype=np.float32)
    for it in range(niter + 1):
        # objective |R CB - X|^2 / n == |CB - R'X|^2 / n for orthonormal R (src/OPQ.jl:108)
        obj[it] = rqd.qerror(RX, CB)
        if V:
            print("%3d %e" % (it, obj[it]))
        # update R: SVD of X CB' (d x d) on the host (src/OPQ.jl:112-113)
        G = rqd.gram(Xd, CB).cpu().numpy().astype(np.float64)    # G[a][b] = sum_j X[j][a] CB[j][b] = (X CB')[a, b]
        U, _, Vt = np.linalg.svd(G, full_matrices=False)
        Rj = U @ Vt                                              # Julia's R (d x d); its memory image is Rj'
        R = np.ascontiguousarray(Rj.T.astype(np.float32))
        Rd = torch.from_numpy(R).to(device)
        rqd.rotate_T(Rd, Xd, out=RX)                             # RX = R' X
        rqd.update_centers(Ccat, RX, codes, m, h)                # update C
        rqd.encode_pq(RX, Ccat, m, h, out=codes)                 # update B
        rqd.reconstruct(codes, Ccat, d, h, out=CB)               # update CB
    B = (codes.cpu().numpy().astype(np.int16) + 1)
    return _split(Ccat, off, h), B, R, obj
