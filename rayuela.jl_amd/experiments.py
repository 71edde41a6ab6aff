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
