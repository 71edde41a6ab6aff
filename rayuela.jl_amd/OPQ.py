"""Host mirror of src/OPQ.jl (encode only)."""
import numpy as np

from . import _lib
from .utils import _as_f32, cat_codebooks


def quantize_opq(X, R, C, V=False):
    """quantize_opq(X, R, C, V=false) -> B      (src/OPQ.jl:19-27) == quantize_pq(R' * X, C, V)

    R : (d, d) float32, the memory image of Julia's d-by-d rotation (so R_numpy[i, k] == R_julia[k, i]).
    Returns (n, m) int16 ONE-based codes.
    """
    X = _as_f32(X, "X")
    R = _as_f32(R, "R")
    n, d = X.shape
    if R.shape != (d, d):
        raise ValueError("R must be %d x %d" % (d, d))
    m = len(C)
    h = np.asarray(C[0]).shape[0]
    Cc = cat_codebooks(C)
    B = _lib.result_empty((n, m), np.int16)
    _lib.check(_lib.lib().rq_encode_opq_i16(B.ctypes.data, X.ctypes.data, R.ctypes.data, Cc.ctypes.data,
                                            n, d, m, h))
    return B


def rotate(R, X):
    """R' * X (src/OPQ.jl:26) in the memory-image convention: RX[j, i] = sum_k R[i, k] X[j, k]."""
    X = _as_f32(X, "X")
    R = _as_f32(R, "R")
    n, d = X.shape
    RX = np.empty((n, d), dtype=np.float32)
    _lib.check(_lib.lib().rq_rotate_T(RX.ctypes.data, R.ctypes.data, X.ctypes.data, d, n))
    return RX


def train_opq(X, m, h, niter, init, V=False, seed=0, R0=None, C0=None):
    """train_opq(X, m, h, niter, init, V=false) -> C, B, R, obj       (src/OPQ.jl:49-139)

    init: "natural" (R = I) or "random"; R0 / C0 optionally pin the initial rotation / codebooks.
    Returns C (list of (h, sub_i)), B (n, m) int16 one-based, R (d, d) memory image of Julia's R,
    obj (niter+1,) float32 -- the objective before every iteration."""
    from .PQ import _split_codebooks
    X = _as_f32(X, "X")
    n, d = X.shape
    if init not in ("natural", "random"):
        raise ValueError("Intialization %s unknown" % init)          # src/OPQ.jl:74
    Ccat = np.empty(h * d, dtype=np.float32)
    B = np.empty((n, m), dtype=np.int16)
    R = np.empty((d, d), dtype=np.float32)
    obj = np.zeros(niter + 1, dtype=np.float32)
    r0 = None if R0 is None else _as_f32(R0, "R0")
    c0 = None if C0 is None else cat_codebooks(C0)
    _lib.check(_lib.lib().rq_train_opq(Ccat.ctypes.data, B.ctypes.data, R.ctypes.data, obj.ctypes.data, X.ctypes.data,
                                       n, d, m, h, niter, 0 if init == "natural" else 1, seed,
                                       None if r0 is None else r0.ctypes.data, None if c0 is None else c0.ctypes.data))
    if V:
        for it, o in enumerate(obj):
            print("%3d %e" % (it, o))
    return _split_codebooks(Ccat, d, m, h), B, R, obj
