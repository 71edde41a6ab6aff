"""Host mirror of src/RVQ.jl (encode): quantize_rvq."""
import numpy as np

from . import _lib
from .utils import _as_f32


def _stack_codebooks(C):
    """C: m-long list of (h, d) float32 arrays (memory image of Julia's d-by-h matrices) -> (m, h, d)."""
    Cs = [np.ascontiguousarray(_as_f32(c, "C[i]")) for c in C]
    if len({c.shape for c in Cs}) != 1:
        raise ValueError("all RVQ codebooks must be d x h")
    return np.ascontiguousarray(np.stack(Cs, axis=0))


def quantize_rvq(X, C, V=False, rng=None):
    """quantize_rvq(X, C, V=false) -> B, singletons          (src/RVQ.jl:18-66)

    X (n, d) float32 (memory image of the d-by-n matrix), C m-long list of (h, d) codebooks.
    Returns B (n, m) int16 ONE-based and `singletons`, an m-long list: entry i is None when every
    centre of codebook i was used, else an (n_unused, d) array of re-picked entries.  The reference
    re-picks them with Clustering.repick_unused_centers driven by Julia's global RNG (:50-53), so those
    VALUES are RNG-specific there as well; here they are drawn with `rng` (numpy Generator) by the same
    rule (a data point sampled with probability proportional to its cost).
    """
    X = _as_f32(X, "X")
    n, d = X.shape
    Cs = _stack_codebooks(C)
    m, h, d2 = Cs.shape
    if d2 != d:
        raise ValueError("codebooks are %d-dimensional, data is %d-dimensional" % (d2, d))
    B = np.empty((n, m), dtype=np.int16)
    counts = np.zeros((m, h), dtype=np.uint32)
    _lib.check(_lib.lib().rq_encode_rvq_i16(B.ctypes.data, X.ctypes.data, Cs.ctypes.data, n, d, m, h,
                                            counts.ctypes.data, None))
    singletons = [None] * m
    if (counts == 0).any():
        rng = np.random.default_rng(0) if rng is None else rng
        Xr = X.astype(np.float32, copy=True)
        for i in range(m):
            unused = np.flatnonzero(counts[i] == 0)
            picked = Cs[i][B[:, i].astype(np.int64) - 1]
            if unused.size:
                costs = ((Xr - picked).astype(np.float64) ** 2).sum(axis=1)
                tot = costs.sum()
                prob = costs / tot if tot > 0 else np.full(n, 1.0 / n)
                singletons[i] = Xr[rng.choice(n, size=unused.size, replace=True, p=prob)].copy()
            Xr -= picked
            if V:
                print("RVQ encoding on codebook %d / %d... done" % (i + 1, m))
    return B, singletons


def quantize_rvq_u8(X, C, with_extras=False):
    """Zero-based uint8 codes (the scan's wire format); with_extras -> (codes, counts, final residual)."""
    X = _as_f32(X, "X")
    n, d = X.shape
    Cs = _stack_codebooks(C)
    m, h, _ = Cs.shape
    B = np.empty((n, m), dtype=np.uint8)
    counts = np.zeros((m, h), dtype=np.uint32)
    Xr = np.empty((n, d), dtype=np.float32) if with_extras else None
    _lib.check(_lib.lib().rq_encode_rvq(B.ctypes.data, X.ctypes.data, Cs.ctypes.data, n, d, m, h,
                                        counts.ctypes.data, None if Xr is None else Xr.ctypes.data))
    return (B, counts, Xr) if with_extras else B


def train_rvq(X, m, h, niter=25, V=False, seed=0):
    """train_rvq(X, m, h, niter=25, V=false) -> C, B, error        (src/RVQ.jl:86-127)

    One k-means per stage on the running residual, all on the device (rq_train_rvq).  C: m-long list of
    (h, d) codebooks; B: (n, m) int16 one-based, equal to quantize_rvq(X, C)[0]; error = qerror(X, B, C).
    Seeding comes from the library's seeded stream (the reference: kmeans++ with Julia's RNG)."""
    import ctypes
    X = _as_f32(X, "X")
    n, d = X.shape
    C = np.empty((m, h, d), dtype=np.float32)
    B = np.empty((n, m), dtype=np.int16)
    err = ctypes.c_double(0.0)
    _lib.check(_lib.lib().rq_train_rvq(C.ctypes.data, B.ctypes.data, ctypes.cast(ctypes.byref(err), ctypes.c_void_p),
                                       X.ctypes.data, n, d, m, h, niter, seed))
    if V:
        print("  Error after codebook %d is %e" % (m, err.value))
    return [C[i] for i in range(m)], B, float(err.value)
