"""Second encode oracle (TEST INFRASTRUCTURE -- never imported by the product): the arithmetic of
`Distances.pairwise(SqEuclidean(), C[i], Xs)` + `Clustering.update_assignments!` (call sites src/PQ.jl:40-41,
src/OPQ.jl:26,100-101,124-125) with a REAL OpenBLAS underneath, the library the reference's `mul!` and `dot` land in.

oracle/rq_oracle.c fixes ONE summation order (k-ordered fmaf chains) because the reference's order lives inside OpenBLAS
and is build- and CPU-specific.  This module keeps the published algorithm

    r   = sgemm('T','N', C_i, Xs)                      # mul!(r, a', b)
    sa2 = sdot(C_i[:,k], C_i[:,k]),  sb2 = sdot(Xs[:,j], Xs[:,j])   # sumsq_percol
    d   = max((sa2[k] + sb2[j]) - 2 r[k,j], 0)         # f32, one rounding per operation
    code[j] = first k with the smallest d               # strict '<' scan

but takes r, sa2, sb2 from the OpenBLAS that ships inside numpy / scipy (`scipy.linalg.blas.sgemm` / `sdot`, same
transposition flags and leading dimensions as the Julia call).  It is NOT the Julia stack (other OpenBLAS build, other
kernels per CPU), so it pins nothing; it measures how far a real BLAS order moves the codes away from the canonical
chain: tests/test_oracle_blas_order.py (CPU, goldens) and tests/test_gpu_encode_flips.py (HIP output at full size)
require every differing code to be a float64 near-tie and print the count."""
import numpy as np
from scipy.linalg import blas as _blas


def blas_version():
    try:
        from threadpoolctl import threadpool_info
        for lib in threadpool_info():
            if lib.get("internal_api") == "openblas":
                return "OpenBLAS %s (%s)" % (lib.get("version"), lib.get("architecture"))
    except Exception:     # noqa: BLE001
        pass
    return "unknown BLAS"


def sumsq_percol(A_rows):
    """A_rows [n][sub] (C view of the Julia sub x n matrix) -> r[j] = dot(view(a,:,j), view(a,:,j)) via BLAS sdot."""
    A_rows = np.ascontiguousarray(A_rows, dtype=np.float32)
    out = np.empty(A_rows.shape[0], dtype=np.float32)
    sdot = _blas.sdot
    for j in range(A_rows.shape[0]):
        out[j] = sdot(A_rows[j], A_rows[j])
    return out


def pairwise_sqeuclidean(Ci, Xs):
    """Ci [h][sub], Xs [n][sub] f32 -> dmat [n][h] (C view of the Julia h x n matrix)."""
    Ci = np.ascontiguousarray(Ci, dtype=np.float32)
    Xs = np.ascontiguousarray(Xs, dtype=np.float32)
    # Ci.T / Xs.T are the Fortran-ordered sub x h / sub x n matrices Julia holds; trans_a=1 is mul!(r, a', b)
    r = _blas.sgemm(np.float32(1.0), Ci.T, Xs.T, trans_a=1)          # h x n, Fortran order
    sa2 = sumsq_percol(Ci)
    sb2 = sumsq_percol(Xs)
    rT = r.T                                                        # [n][h] C view
    s = sa2[None, :] + sb2[:, None]                                 # fl(sa2[i] + sb)
    d = s - np.float32(2.0) * rT                                    # 2r is exact
    np.maximum(d, np.float32(0.0), out=d)
    return d


def encode_pq(X, C_list, offsets=None, chunk=65536):
    """codes [n][m] u8 zero-based.  C_list[i] [h][sub_i]; offsets = splitarray boundaries (src/utils.jl:179-203)."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    n, d = X.shape
    m = len(C_list)
    if offsets is None:
        offsets = np.concatenate([[0], np.cumsum([c.shape[1] for c in C_list])])
    assert offsets[-1] == d
    codes = np.empty((n, m), dtype=np.uint8)
    for i, Ci in enumerate(C_list):
        for a in range(0, n, chunk):
            dm = pairwise_sqeuclidean(Ci, X[a:a + chunk, offsets[i]:offsets[i + 1]])
            codes[a:a + chunk, i] = np.argmin(dm, axis=1)           # first index of the minimum == strict '<' scan
    return codes


def rotate_T(R, X):
    """RX = R'X (src/OPQ.jl:26) through sgemm('T','N'): C views X [n][d], R [d][d] with Rc[i][k] = R[k,i]."""
    R = np.ascontiguousarray(R, dtype=np.float32)
    X = np.ascontiguousarray(X, dtype=np.float32)
    # Julia's R (column-major d x d) has R[k,i] at k + i d, i.e. our Rc[i][k]: the Fortran view of Rc is Rc.T
    out = _blas.sgemm(np.float32(1.0), R.T, X.T, trans_a=1)         # (R')X, d x n Fortran order
    return np.ascontiguousarray(out.T)


def near_tie_report(X, C_list, offsets, codes_a, codes_b):
    """Every position where the two code arrays differ, judged in float64: (count, count outside the f32 rounding bound,
    worst gap / bound).  The bound is the one of tests/test_gpu_encode_flips.py."""
    X = np.asarray(X, dtype=np.float64)
    flips = outside = 0
    worst = 0.0
    for i, Ci in enumerate(C_list):
        rows = np.nonzero(codes_a[:, i] != codes_b[:, i])[0]
        if rows.size == 0:
            continue
        C64 = np.asarray(Ci, dtype=np.float64)
        sub = C64.shape[1]
        cc = (C64 * C64).sum(1)
        Xs = X[rows, offsets[i]:offsets[i + 1]]
        xx = (Xs * Xs).sum(1)
        d64 = np.maximum(xx[:, None] + cc[None, :] - 2.0 * Xs @ C64.T, 0.0)
        da = d64[np.arange(rows.size), codes_a[rows, i]]
        db = d64[np.arange(rows.size), codes_b[rows, i]]
        gap = np.abs(da - db)
        bound = 2.0 * (sub + 2) * 2.0 ** -24 * (2.0 * xx + cc[codes_a[rows, i]] + cc[codes_b[rows, i]])   # the two centroids involved
        flips += rows.size
        outside += int((gap > bound).sum())
        worst = max(worst, float((gap / bound).max()))
    return flips, outside, worst
