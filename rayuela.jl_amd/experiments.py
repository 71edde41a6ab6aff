"""Host mirror of the end-to-end drivers: experiment_pq (src/PQ.jl:104-132), experiment_pq_query_base
(:137-159), experiment_opq (src/OPQ.jl:142-171), experiment_opq_query_base (:174-197).
train -> encode the base -> ADC search -> recall, every O(n) step on the device."""
import numpy as np

from .Linscan import eval_recall, linscan_opq, linscan_pq
from .OPQ import quantize_opq, train_opq
from .PQ import quantize_pq, train_pq


def _qerror(X, B, C, R=None):
    """qerror_pq / qerror_opq (src/qerrors.jl:77-100): mean squared reconstruction error."""
    n, d = X.shape
    CB = np.concatenate([np.asarray(C[i])[B[:, i].astype(np.int64) - 1] for i in range(len(C))], axis=1)
    RX = X if R is None else X.astype(np.float64) @ np.asarray(R, dtype=np.float64).T   # R'X in memory-image form
    return float(((RX.astype(np.float64) - CB) ** 2).sum() / n)


def experiment_pq(Xt, Xb, Xq, gt, m, h, niter=25, knn=1000, V=False, seed=0):
    C, B, train_error = train_pq(Xt, m, h, niter, V, seed=seed)
    if V:
        print("Error in training is %e" % train_error)
    B_base = quantize_pq(Xb, C, V)
    base_error = _qerror(Xb, B_base, C)
    if V:
        print("Error in base is %e" % base_error)
    b = int(np.log2(h) * m)
    dists, idx = linscan_pq(B_base, Xq, C, b, knn)
    recall = eval_recall(gt, idx, knn, verbose=V)
    return C, B, train_error, B_base, recall


def experiment_pq_query_base(Xt, Xq, gt, m, h, niter=25, knn=1000, V=False, seed=0):
    C, B, train_error = train_pq(Xt, m, h, niter, V, seed=seed)
    b = int(np.log2(h) * m)
    dists, idx = linscan_pq(B, Xq, C, b, knn)
    recall = eval_recall(gt, idx, knn, verbose=V)
    return C, B, train_error, recall


def experiment_opq(Xt, Xb, Xq, gt, m, h, init, niter=25, knn=1000, V=False, seed=0):
    C, B, R, train_error = train_opq(Xt, m, h, niter, init, V, seed=seed)
    if V:
        print("Error in training is %e" % train_error[-1])
    B_base = quantize_opq(Xb, R, C, V)
    base_error = _qerror(Xb, B_base, C, R)
    if V:
        print("Error in base is %e" % base_error)
    b = int(np.log2(h) * m)
    dists, idx = linscan_opq(B_base, Xq, C, b, R, knn)
    recall = eval_recall(gt, idx, knn, verbose=V)
    return C, B, R, train_error, B_base, recall


def experiment_opq_query_base(Xt, Xq, gt, m, h, init, niter=25, knn=1000, V=False, seed=0):
    C, B, R, train_error = train_opq(Xt, m, h, niter, init, V, seed=seed)
    b = int(np.log2(h) * m)
    dists, idx = linscan_opq(B, Xq, C, b, R, knn)
    recall = eval_recall(gt, idx, knn, verbose=V)
    return C, B, R, train_error, recall


def _norms_codebook(B, C, h=256, seed=0):
    """get_norms_codebook (src/utils.jl:4-26): norms of the reconstructions, quantised by a 1-D k-means
    (rq_train_pq with d = m = 1).  Returns (norms_codes one-based, norms_codebook (h,))."""
    from .PQ import train_pq
    Cs = np.stack([np.asarray(c, dtype=np.float32) for c in C])
    codes = B.astype(np.int64) - 1
    recon = np.zeros((B.shape[0], Cs.shape[2]), dtype=np.float32)
    for i in range(Cs.shape[0]):
        recon += Cs[i][codes[:, i]]
    dbnorms = (recon ** 2).sum(axis=1, dtype=np.float32).reshape(-1, 1)
    Cn, Bn, _ = train_pq(np.ascontiguousarray(dbnorms), 1, h, niter=25, seed=seed)
    return Bn[:, 0].astype(np.int64), Cn[0][:, 0].copy()


def _quantize_norms(B, C, norms_C):
    """quantize_norms (src/utils.jl:29-60): nearest entry of the norms codebook for every reconstruction."""
    Cs = np.stack([np.asarray(c, dtype=np.float32) for c in C])
    codes = B.astype(np.int64) - 1
    recon = np.zeros((B.shape[0], Cs.shape[2]), dtype=np.float32)
    for i in range(Cs.shape[0]):
        recon += Cs[i][codes[:, i]]
    dbnorms = (recon ** 2).sum(axis=1, dtype=np.float32)
    order = np.argsort(norms_C, kind="stable")
    srt = norms_C[order]
    pos = np.clip(np.searchsorted(srt, dbnorms), 1, len(srt) - 1)
    left = np.abs(dbnorms - srt[pos - 1]) <= np.abs(dbnorms - srt[pos])
    return order[np.where(left, pos - 1, pos)] + 1, dbnorms


def experiment_rvq(Xt, Xb, Xq, gt, m, h, niter=25, knn=1000, V=False, seed=0):
    """experiment_rvq (src/RVQ.jl:130-175): train_rvq -> norms codebook -> quantize_rvq of the base ->
    quantised database norms -> linscan_lsq -> eval_recall."""
    from .RVQ import train_rvq, quantize_rvq
    from .Linscan import linscan_lsq
    d = Xt.shape[1]
    C, B, train_error = train_rvq(Xt, m, h, niter, V, seed=seed)
    _, norms_C = _norms_codebook(B, C, h, seed=seed)
    B_base, _ = quantize_rvq(Xb, C, V)
    B_base_norms, _ = _quantize_norms(B_base, C, norms_C)
    db_norms = norms_C[B_base_norms - 1].astype(np.float32)
    dists, idx = linscan_lsq(B_base, Xq, C, db_norms, np.eye(d, dtype=np.float32), knn)
    recall = eval_recall(gt, idx, knn, verbose=V)
    return C, B, train_error, B_base, recall
