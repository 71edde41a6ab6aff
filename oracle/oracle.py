"""ctypes front-end of the parity oracle (TEST INFRASTRUCTURE -- never imported by the product).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  It wraps

  * oracle/liboracle.so        -- our C restatement (oracle/rq_oracle.c), and
  * oracle/_ref/linscan_aqd.so -- the real reference scan compiled from
                                  /root/reference/deps/src/linscan_aqd.cpp by oracle/Makefile
                                  (present only if it was built in the build container).

All arrays are numpy, C-contiguous, in the C views of the Julia layouts:
  X [n][d] f32, centers [m][256][sub] f32 (== cat(C...,dims=3)), codes [n][m] u8 zero-based,
  queries [nq][d] f32, outputs [nq][K].
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)


def build(with_ref=True):
    """Compile liboracle.so (always) and oracle/_ref (only when /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if with_ref and os.path.isfile("/root/reference/deps/src/linscan_aqd.cpp"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.isfile(path):
            build(with_ref=False)
        _LIB = C.CDLL(path)
        _LIB.oracle_linscan_aqd_query.restype = None
        _LIB.oracle_linscan_aqd_query.argtypes = [
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
            C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        _LIB.oracle_adc_lut.restype = None
        _LIB.oracle_adc_lut.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        _LIB.oracle_adc_distances.restype = None
        _LIB.oracle_adc_distances.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_int, C.c_int, C.c_int]
        _LIB.oracle_rotate_T.restype = None
        _LIB.oracle_rotate_T.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64]
        _LIB.oracle_encode_pq.restype = None
        _LIB.oracle_encode_pq.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_int64, C.c_int, C.c_int, C.c_int]
        _LIB.oracle_pq_distmat.restype = None
        _LIB.oracle_pq_distmat.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int]
        _LIB.oracle_encode_opq.restype = None
        _LIB.oracle_encode_opq.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int64, C.c_int, C.c_int, C.c_int]
        _LIB.oracle_encode_rvq.restype = None
        _LIB.oracle_encode_rvq.argtypes = [C.c_void_p] * 5 + [C.c_int64, C.c_int, C.c_int, C.c_int]
        _LIB.oracle_splitarray.restype = None
        _LIB.oracle_splitarray.argtypes = [C.c_int, C.c_int, C.c_void_p]
        _LIB.oracle_num_threads.restype = C.c_int
        _LIB.oracle_linscan_aq.restype = None
        _LIB.oracle_linscan_aq.argtypes = [C.c_void_p] * 6 + [C.c_int] * 7
    return _LIB


def ref_available():
    return os.path.isfile(os.path.join(_HERE, "_ref", "linscan_aqd.so"))


def ref():
    """The compiled reference linscan_aqd.so (deps/src/linscan_aqd.cpp:105-114)."""
    global _REF
    if _REF is None:
        _REF = C.CDLL(os.path.join(_HERE, "_ref", "linscan_aqd.so"))
        _REF.linscan_aqd_query.restype = None
        _REF.linscan_aqd_query.argtypes = [
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
            C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    return _REF


_REF_AQ = None


def ref_aq_available():
    return os.path.isfile(os.path.join(_HERE, "_ref", "linscan_aqd_pairwise_byte.so"))


def ref_aq():
    """The compiled reference linscan_aqd_pairwise_byte.so (deps/src/linscan_aqd_pairwise_byte.cpp:179-198)."""
    global _REF_AQ
    if _REF_AQ is None:
        _REF_AQ = C.CDLL(os.path.join(_HERE, "_ref", "linscan_aqd_pairwise_byte.so"))
        _REF_AQ.linscan_aqd_query_extra_byte.restype = None
        _REF_AQ.linscan_aqd_query_extra_byte.argtypes = [C.c_void_p] * 6 + [C.c_int] * 6
        _REF_AQ.linscan_aqd_cq_query_extra_byte.restype = None
        _REF_AQ.linscan_aqd_cq_query_extra_byte.argtypes = [C.c_void_p] * 5 + [C.c_int] * 6
    return _REF_AQ


def _aq_args(codes, codebooks, queries, dbnorms):
    codes = _c(codes, np.uint8)
    codebooks = _c(codebooks, np.float32)     # [m*h][d]
    queries = _c(queries, np.float32)
    n, m = codes.shape
    nq, d = queries.shape
    h = codebooks.shape[0] // m
    assert codebooks.shape == (m * h, d)
    if dbnorms is not None:
        dbnorms = _c(dbnorms, np.float32)
        assert dbnorms.shape == (n,)
    return codes, codebooks, queries, dbnorms, n, m, nq, d, h


def linscan_lsq(codes, codebooks, queries, dbnorms, K, use_ref=False):
    """LSQ scan (restatement, or the real reference with use_ref).  ids ONE-based int32."""
    codes, codebooks, queries, dbnorms, n, m, nq, d, h = _aq_args(codes, codebooks, queries, dbnorms)
    dists = np.zeros((nq, K), dtype=np.float32)
    idx = np.zeros((nq, K), dtype=np.int32)
    if use_ref:
        ref_aq().linscan_aqd_query_extra_byte(_ptr(dists), _ptr(idx), _ptr(codes), _ptr(queries), _ptr(codebooks),
                                              _ptr(dbnorms), nq, n, m, h, d, K)
    else:
        lib().oracle_linscan_aq(_ptr(dists), _ptr(idx), _ptr(codes), _ptr(queries), _ptr(codebooks), _ptr(dbnorms),
                                nq, n, m, h, d, K, 1)
    return dists, idx


def linscan_cq(codes, codebooks, queries, K, use_ref=False):
    codes, codebooks, queries, _, n, m, nq, d, h = _aq_args(codes, codebooks, queries, None)
    dists = np.zeros((nq, K), dtype=np.float32)
    idx = np.zeros((nq, K), dtype=np.int32)
    if use_ref:
        ref_aq().linscan_aqd_cq_query_extra_byte(_ptr(dists), _ptr(idx), _ptr(codes), _ptr(queries), _ptr(codebooks),
                                                 nq, n, m, h, d, K)
    else:
        lib().oracle_linscan_aq(_ptr(dists), _ptr(idx), _ptr(codes), _ptr(queries), _ptr(codebooks), None,
                                nq, n, m, h, d, K, 2)
    return dists, idx


def _c(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    return a


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def num_threads():
    return int(lib().oracle_num_threads())


def splitarray(d, m):
    off = np.zeros(m + 1, dtype=np.int32)
    lib().oracle_splitarray(d, m, _ptr(off))
    return off


def _scan(fn, codes, centers, queries, K):
    codes = _c(codes, np.uint8)
    centers = _c(centers, np.float32)
    queries = _c(queries, np.float32)
    n, m = codes.shape
    nq, d = queries.shape
    assert centers.shape[0] == m and centers.shape[1] == 256
    sub = centers.shape[2]
    assert sub * m == d, "scan needs d % m == 0 (src/Linscan.jl:23)"
    assert 1 <= K <= n
    dists = np.zeros((nq, K), dtype=np.float32)
    ids = np.zeros((nq, K), dtype=np.uint32)
    fn(_ptr(dists), _ptr(ids), _ptr(codes), _ptr(centers), _ptr(queries),
       n, nq, 8 * m, K, m, d, sub)
    return dists, ids


def linscan_aqd_query(codes, centers, queries, K):
    """Our restatement.  Returns (dists [nq][K] f32, ids [nq][K] u32 zero-based)."""
    return _scan(lib().oracle_linscan_aqd_query, codes, centers, queries, K)


def ref_linscan_aqd_query(codes, centers, queries, K):
    """The real reference (needs oracle/_ref/linscan_aqd.so)."""
    return _scan(ref().linscan_aqd_query, codes, centers, queries, K)


def adc_lut(centers, query):
    centers = _c(centers, np.float32)
    query = _c(query, np.float32)
    m, h, sub = centers.shape
    assert h == 256
    lut = np.zeros((m, 256), dtype=np.float32)
    lib().oracle_adc_lut(_ptr(lut), _ptr(centers), _ptr(query), m, sub)
    return lut


def adc_distances(codes, centers, query):
    codes = _c(codes, np.uint8)
    centers = _c(centers, np.float32)
    query = _c(query, np.float32)
    n, m = codes.shape
    out = np.zeros(n, dtype=np.float32)
    lib().oracle_adc_distances(_ptr(out), _ptr(codes), _ptr(centers), _ptr(query),
                               n, m, centers.shape[2])
    return out


def rotate_T(R, X):
    """RX = R'X in the C views: X [n][d], R [d][d] (Rc[i][k] = R[k,i]), RX [n][d]."""
    R = _c(R, np.float32)
    X = _c(X, np.float32)
    n, d = X.shape
    RX = np.zeros((n, d), dtype=np.float32)
    lib().oracle_rotate_T(_ptr(RX), _ptr(R), _ptr(X), d, n)
    return RX


def encode_pq(X, C_cat, m, h, with_costs=False):
    """codes [n][m] u8 zero-based.  C_cat: flat concat over subspaces of [h][sub_i] blocks
    (for even splits simply centers [m][h][sub])."""
    X = _c(X, np.float32)
    Cc = _c(np.asarray(C_cat).reshape(-1), np.float32)
    n, d = X.shape
    assert Cc.size == h * d
    codes = np.zeros((n, m), dtype=np.uint8)
    costs = np.zeros((n, m), dtype=np.float32) if with_costs else None
    lib().oracle_encode_pq(_ptr(codes), _ptr(costs) if with_costs else None, _ptr(X), _ptr(Cc),
                           n, d, m, h)
    return (codes, costs) if with_costs else codes


def pq_distmat(X, C_cat, m, h):
    """U [n][m][h]: the unclamped canonical distances fl(fl(sa + sb) - 2 g) of quantize_pq's dmat (even splits)."""
    X = _c(X, np.float32)
    Cc = _c(np.asarray(C_cat).reshape(-1), np.float32)
    n, d = X.shape
    assert d % m == 0 and Cc.size == h * d
    U = np.empty((n, m, h), dtype=np.float32)
    lib().oracle_pq_distmat(_ptr(U), _ptr(X), _ptr(Cc), n, d, m, h)
    return U


def encode_opq(X, R, C_cat, m, h):
    X = _c(X, np.float32)
    R = _c(R, np.float32)
    Cc = _c(np.asarray(C_cat).reshape(-1), np.float32)
    n, d = X.shape
    codes = np.zeros((n, m), dtype=np.uint8)
    lib().oracle_encode_opq(_ptr(codes), _ptr(X), _ptr(R), _ptr(Cc), n, d, m, h)
    return codes


def encode_rvq(X, C, with_extras=False):
    """src/RVQ.jl:18-66.  C [m][h][d] full-dimensional codebooks; codes [n][m] u8 zero-based.
    with_extras -> (codes, counts [m][h], final residual [n][d])."""
    X = _c(X, np.float32)
    C = _c(C, np.float32)
    n, d = X.shape
    m, h, d2 = C.shape
    assert d2 == d
    codes = np.zeros((n, m), dtype=np.uint8)
    counts = np.zeros((m, h), dtype=np.uint32)
    Xr = np.zeros((n, d), dtype=np.float32)
    lib().oracle_encode_rvq(_ptr(codes), _ptr(counts), _ptr(Xr), _ptr(X), _ptr(C), n, d, m, h)
    return (codes, counts, Xr) if with_extras else codes


def eval_recall(ids_gnd, ids_predicted, k):
    """src/Linscan.jl:196-234.  ids_gnd [nq], ids_predicted [nq][k] (C view of the k x nq
    Julia matrix), same index base.  rank = position (1-based) of the ground-truth id if it
    occurs EXACTLY once in the WHOLE column of predictions (:207, also beyond k), else k+1;
    recall_at_i[R-1] = #{rank<=R and rank<=k}/nq.  Explicit loops, written from the reference's text."""
    ids_gnd = np.asarray(ids_gnd).reshape(-1)
    P = np.asarray(ids_predicted)
    nq = P.shape[0]
    assert nq == ids_gnd.shape[0]
    hit = P == ids_gnd[:, None]
    cnt = hit.sum(axis=1)
    first = hit.argmax(axis=1) + 1
    ranks = np.minimum(np.where(cnt == 1, first, k + 1), k + 1)
    hist = np.bincount(ranks, minlength=k + 2)[1:k + 1]
    return np.cumsum(hist) / float(nq)
