"""Host mirror of src/Linscan.jl: linscan_pq, linscan_opq, eval_recall."""
import numpy as np

from . import _lib
from .utils import _as_f32


def _codes_u8(B):
    B = np.asarray(B)
    if B.dtype == np.uint8:
        return np.ascontiguousarray(B)  # already the zero-based wire format (src/Linscan.jl:5-10)
    if not np.issubdtype(B.dtype, np.integer):
        raise TypeError("B must be an integer array")
    Bm1 = B.astype(np.int64) - 1  # src/Linscan.jl:35  convert(Matrix{UInt8}, B .- 1)
    if Bm1.min(initial=0) < 0 or Bm1.max(initial=0) > 255:
        raise OverflowError("InexactError: a one-based code outside 1..256 does not fit UInt8")
    return np.ascontiguousarray(Bm1.astype(np.uint8))


def _centers(C, m, d):
    sub = d // m
    arr = np.stack([_as_f32(c, "C[i]") for c in C])
    if arr.shape != (m, 256, sub):
        raise ValueError("linscan needs m codebooks of shape (256, d/m); got %s" % (arr.shape,))
    return np.ascontiguousarray(arr)  # == cat(C..., dims=3), src/Linscan.jl:22


def linscan_pq(B, X, C, b, k=10000):
    """linscan_pq(B, X, C, b, k=10000) -> dists, idx          (src/Linscan.jl:5-37)

    B : (n, m) uint8 zero-based codes, or any other integer dtype holding ONE-based codes
    X : (nq, d) float32 queries;  C : list of m (256, d/m) codebooks;  b = log2(h) * m
    Returns dists (nq, k) float32 ascending and idx (nq, k) uint32, ONE-based like the reference.
    """
    Bu = _codes_u8(B)
    X = _as_f32(X, "X")
    n, m = Bu.shape
    nq, d = X.shape
    if d % m:
        raise ValueError("InexactError: Cint(d/m) with d=%d m=%d (src/Linscan.jl:23)" % (d, m))
    if b != 8 * m:
        raise ValueError("b must be log2(256)*m = %d" % (8 * m))
    cen = _centers(C, m, d)
    dists = _lib.result_empty((nq, k), np.float32)     # every element is written by the library
    idx = _lib.result_empty((nq, k), np.uint32)
    _lib.check(_lib.lib().rq_linscan_pq(dists.ctypes.data, idx.ctypes.data, Bu.ctypes.data, cen.ctypes.data,
                                        X.ctypes.data, n, nq, m, d, k, 1))
    return dists, idx


def linscan_opq(B, X, C, b, R, k=10000):
    """linscan_opq(B, X, C, b, R, k=10000)   (src/Linscan.jl:93-115) == linscan_pq(B, R'X, C, b, k)."""
    Bu = _codes_u8(B)
    X = _as_f32(X, "X")
    R = _as_f32(R, "R")
    n, m = Bu.shape
    nq, d = X.shape
    if d % m:
        raise ValueError("InexactError: Cint(d/m) with d=%d m=%d" % (d, m))
    if b != 8 * m:
        raise ValueError("b must be log2(256)*m = %d" % (8 * m))
    cen = _centers(C, m, d)
    dists = _lib.result_empty((nq, k), np.float32)     # every element is written by the library
    idx = _lib.result_empty((nq, k), np.uint32)
    _lib.check(_lib.lib().rq_linscan_opq(dists.ctypes.data, idx.ctypes.data, Bu.ctypes.data, cen.ctypes.data,
                                         X.ctypes.data, R.ctypes.data, n, nq, m, d, k, 1))
    return dists, idx


def _hcat(C, m, d):
    arr = np.concatenate([_as_f32(c, "C[i]") for c in C], axis=0)   # hcat(C...) == [m*h][d] in memory
    if arr.shape != (m * 256, d):
        raise ValueError("linscan_lsq/cq need m codebooks of shape (256, d); got %s" % (arr.shape,))
    return np.ascontiguousarray(arr)


def linscan_lsq(B, X, C, dbnorms, R, k=10000):
    """linscan_lsq(B, X, C, dbnorms, R, k=10000) -> dists, idx      (src/Linscan.jl:118-157)

    ADC search for additive (non-orthogonal) quantizers with the database norms passed apart:
    table T = -2<R'x, c> per codebook entry, dist = sum_k T[k][b_k] + dbnorms[row].
    B (n, m) uint8 zero-based (or other ints one-based); C list of m (256, d) full-dimensional codebooks;
    dbnorms (n,) float32; R (d, d).  idx ONE-based (the C code already returns them so, :76)."""
    Bu = _codes_u8(B)
    X = _as_f32(X, "X")
    R = _as_f32(R, "R")
    n, m = Bu.shape
    nq, d = X.shape
    cb = _hcat(C, m, d)
    nrm = np.ascontiguousarray(dbnorms, dtype=np.float32)
    if nrm.shape != (n,):
        raise ValueError("dbnorms must have one entry per database row")
    dists = _lib.result_empty((nq, k), np.float32)     # every element is written by the library
    idx = _lib.result_empty((nq, k), np.uint32)
    _lib.check(_lib.lib().rq_linscan_lsq(dists.ctypes.data, idx.ctypes.data, Bu.ctypes.data, X.ctypes.data,
                                         cb.ctypes.data, nrm.ctypes.data, R.ctypes.data, n, nq, m, 256, d, k, 1))
    return dists, idx


class LsqIndex:
    """linscan_lsq over a PREPARED base (rq_lsq_prepare): codes, dbnorms and codebooks stay on the device together with the
    pre-filter's O(n) preprocessing, which linscan_lsq otherwise repeats on every call.  search(X, R, k) returns exactly what
    linscan_lsq(B, X, C, dbnorms, R, k) returns."""

    def __init__(self, B, C, dbnorms):
        Bu = _codes_u8(B)
        n, m = Bu.shape
        d = np.asarray(C[0]).shape[1]
        cb = _hcat(C, m, d)
        nrm = np.ascontiguousarray(dbnorms, dtype=np.float32)
        if nrm.shape != (n,):
            raise ValueError("dbnorms must have one entry per database row")
        self.n, self.m, self.d = n, m, d
        self._h = _lib.lib().rq_lsq_prepare(Bu.ctypes.data, cb.ctypes.data, nrm.ctypes.data, n, m, 256, d)
        if not self._h:
            raise _lib.RayuelaHipError("rq_lsq_prepare: " + _lib.lib().rq_last_error().decode("utf-8", "replace"))

    def search(self, X, R=None, k=10000):
        X = _as_f32(X, "X")
        nq, d = X.shape
        if d != self.d:
            raise ValueError("queries must have d=%d columns" % self.d)
        Rp = None if R is None else _as_f32(R, "R")
        dists = _lib.result_empty((nq, k), np.float32)
        idx = _lib.result_empty((nq, k), np.uint32)
        _lib.check(_lib.lib().rq_lsq_search(self._h, dists.ctypes.data, idx.ctypes.data, X.ctypes.data,
                                            None if Rp is None else Rp.ctypes.data, nq, k, 1))
        return dists, idx

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().rq_lsq_release(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def linscan_cq(B, X, C, k=10000):
    """linscan_cq(B, X, C, k=10000) -> dists, idx      (src/Linscan.jl:160-193): T = |x - c|^2 per entry."""
    Bu = _codes_u8(B)
    X = _as_f32(X, "X")
    n, m = Bu.shape
    nq, d = X.shape
    cb = _hcat(C, m, d)
    dists = _lib.result_empty((nq, k), np.float32)     # every element is written by the library
    idx = _lib.result_empty((nq, k), np.uint32)
    _lib.check(_lib.lib().rq_linscan_cq(dists.ctypes.data, idx.ctypes.data, Bu.ctypes.data, X.ctypes.data,
                                        cb.ctypes.data, n, nq, m, 256, d, k, 1))
    return dists, idx


def linscan_aqd_query_extra_byte(codes, queries, codebooks, dbnorms, K):
    """Raw C symbol of deps/src/linscan_aqd_pairwise_byte.cpp:181-188 (ids ONE-based int32)."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    codebooks = np.ascontiguousarray(codebooks, dtype=np.float32)
    n, m = codes.shape
    nq, d = queries.shape
    h = codebooks.shape[0] // m
    dists = np.zeros((nq, K), dtype=np.float32)
    idx = np.zeros((nq, K), dtype=np.int32)
    if dbnorms is None:
        _lib.lib().linscan_aqd_cq_query_extra_byte(dists.ctypes.data, idx.ctypes.data, codes.ctypes.data,
                                                   queries.ctypes.data, codebooks.ctypes.data, nq, n, m, h, d, K)
    else:
        dbnorms = np.ascontiguousarray(dbnorms, dtype=np.float32)
        _lib.lib().linscan_aqd_query_extra_byte(dists.ctypes.data, idx.ctypes.data, codes.ctypes.data,
                                                queries.ctypes.data, codebooks.ctypes.data, dbnorms.ctypes.data,
                                                nq, n, m, h, d, K)
    return dists, idx


def linscan_aqd_query(codes, centers, queries, K):
    """The raw C symbol of deps/src/linscan_aqd.cpp:105-114 (zero-based ids), as Linscan.jl ccalls it."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    centers = np.ascontiguousarray(centers, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    n, m = codes.shape
    nq, d = queries.shape
    dists = np.zeros((nq, K), dtype=np.float32)
    res = np.zeros((nq, K), dtype=np.uint32)
    _lib.lib().linscan_aqd_query(dists.ctypes.data, res.ctypes.data, codes.ctypes.data, centers.ctypes.data,
                                 queries.ctypes.data, n, nq, 8 * m, K, m, d, d // m)
    return dists, res


def eval_recall(ids_gnd, ids_predicted, k, verbose=True):
    """eval_recall(gt, idx, k) -> recall_at_i     (src/Linscan.jl:196-234)

    rank_i = position of gt[i] in idx[i, :] (the whole row, like the reference) if it occurs exactly once, else k+1;
    recall_at_i[R-1] = #{rank <= R} / nq.  Prints r@{1,2,5,...} * 100 like the reference."""
    ids_gnd = np.asarray(ids_gnd).reshape(-1)
    P = np.asarray(ids_predicted)             # the WHOLE column is searched (src/Linscan.jl:207), also beyond k
    nq = P.shape[0]
    assert nq == ids_gnd.shape[0]
    hit = P == ids_gnd[:, None]
    cnt = hit.sum(axis=1)
    ranks = np.where(cnt == 1, hit.argmax(axis=1) + 1, k + 1)        # :209-213: exactly one occurrence, else k+1
    ranks = np.minimum(ranks, k + 1)                                  # a rank beyond k counts for no R <= k (:228)
    hist = np.bincount(ranks, minlength=k + 2)[1:k + 1]
    recall = np.cumsum(hist) / float(nq)
    if verbose:
        for i in (1, 2, 5, 10, 20, 50, 100, 200, 500, 1000, 2000, 5000, 10000):
            if i <= k:
                print("r@%d = %s" % (i, recall[i - 1] * 100))
    return recall
