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
    B = np.empty((n, m), dtype=np.int16)
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
