"""Host mirror of src/PQ.jl (encode only): same name, argument order and return convention."""
import numpy as np

from . import _lib
from .utils import _as_f32, cat_codebooks


def quantize_pq(X, C, V=False):
    """quantize_pq(X, C, V=false) -> B          (src/PQ.jl:18-48)

    X : (n, d) float32 -- the memory image of Julia's d-by-n matrix
    C : list of m arrays (h, sub_i) float32 -- memory images of the sub_i-by-h codebooks
    Returns B : (n, m) int16, ONE-based codes (memory image of Julia's m-by-n Matrix{Int16}).
    """
    X = _as_f32(X, "X")
    n, d = X.shape
    m = len(C)
    h = np.asarray(C[0]).shape[0]
    Cc = cat_codebooks(C)
    if Cc.size != h * d:
        raise ValueError("codebooks do not tile the %d dimensions of X" % d)
    if V:
        print("Encoding on %d codebooks with librayuela_hip... " % m, end="")
    B = _lib.result_empty((n, m), np.int16)
    _lib.check(_lib.lib().rq_encode_pq_i16(B.ctypes.data, X.ctypes.data, Cc.ctypes.data, n, d, m, h))
    if V:
        print("done")
    return B


def quantize_pq_u8(X, C):
    """Same encode, returning the zero-based uint8 wire format the scan consumes
    (convert(Matrix{UInt8}, B .- 1), src/Linscan.jl:35)."""
    X = _as_f32(X, "X")
    n, d = X.shape
    m = len(C)
    h = np.asarray(C[0]).shape[0]
    Cc = cat_codebooks(C)
    B = _lib.result_empty((n, m), np.uint8)
    _lib.check(_lib.lib().rq_encode_pq(B.ctypes.data, X.ctypes.data, Cc.ctypes.data, n, d, m, h))
    return B


def train_pq(X, m, h, niter=25, V=False, seed=0):
    """train_pq(X, m, h, niter=25, V=false) -> C, B, error        (src/PQ.jl:68-99)

    Lloyd's k-means in every subspace (all m subspaces per device pass), through rq_train_pq.
    C: list of m (h, sub_i) codebooks; B: (n, m) int16 one-based; error: qerror_pq of the result.
    Initial centres: kmeans++ like the reference (kmeans(..., init=:kmpp)), drawn from the library's seeded
    stream rather than Julia's RNG (see include/rayuela_hip.h)."""
    X = _as_f32(X, "X")
    n, d = X.shape
    Ccat = np.empty(h * d, dtype=np.float32)
    B = np.empty((n, m), dtype=np.int16)
    import ctypes
    err = ctypes.c_double(0.0)
    _lib.check(_lib.lib().rq_train_pq(Ccat.ctypes.data, B.ctypes.data, ctypes.cast(ctypes.byref(err), ctypes.c_void_p),
                                      X.ctypes.data, n, d, m, h, niter, seed))
    if V:
        print("  Error in training is %e" % err.value)
    return _split_codebooks(Ccat, d, m, h), B, float(err.value)


def kmpp_seeds(X, m, h, seed=0):
    """kmeans++ seeding of the m sub-spaces as train_pq uses it (Clustering.jl init=:kmpp, src/PQ.jl:86):
    returns (seeds (m, h) int64 zero-based rows of X, C list of m (h, sub_i) seed sub-vectors)."""
    X = _as_f32(X, "X")
    n, d = X.shape
    seeds = np.empty((m, h), dtype=np.int64)
    Ccat = np.empty(h * d, dtype=np.float32)
    _lib.check(_lib.lib().rq_kmpp_seeds(seeds.ctypes.data, Ccat.ctypes.data, X.ctypes.data, n, d, m, h, seed))
    return seeds, _split_codebooks(Ccat, d, m, h)


def _split_codebooks(Ccat, d, m, h):
    per, extra = divmod(d, m)
    out, pos = [], 0
    for i in range(m):
        sub = per + (1 if i < extra else 0)
        out.append(Ccat[pos:pos + h * sub].reshape(h, sub).copy())
        pos += h * sub
    return out
