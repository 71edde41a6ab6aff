"""The ADC scan on non-finite inputs (VERDICT r5 Missing #5).  The reference builds the table and the distances in plain f32
(deps/src/linscan_aqd.cpp:66-87) and hands whatever comes out to std::partial_sort on (float, id) pairs (:91-97): +Inf is an
ordinary value there (ties part by id); a NaN makes the pair comparison no strict weak order, so the reference's answer for
that query is unspecified.  What the HIP path guarantees (include/rayuela_hip.h, "Non-finite inputs"):
  * the call returns (no redo / barrier / pacing path spins on a poisoned group);
  * every query without a NaN distance -- the other queries of the same group and of the same launch -- is bit-identical to the
    reference, +Inf / -Inf distances included;
  * a row whose distance to a query is NaN is never a neighbour of that query; the rows with comparable distances are returned
    exactly, in (dist, id) order, and a list that runs out of them ends in (NaN, id 0xFFFFFFFF zero-based = 0 one-based).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _eq_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def _case(seed, n, m, sub, nq):
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((m, 256, sub)).astype(np.float32)
    queries = rng.standard_normal((nq, m * sub)).astype(np.float32)
    codes = rng.integers(0, 256, (n, m), dtype=np.uint8)
    return centers, queries, codes


SHAPES = [
    (300_000, 8, 4, 29, 100),      # sampled threshold + byte pre-filter + second estimate, ragged last group
    (300_000, 8, 4, 29, 1000),
    (120_000, 16, 2, 21, 100),     # m = 16: the 1024-thread kernel
    (40_000, 8, 4, 13, 4096),      # large k: sample-sort finish
    (3_000, 8, 4, 13, 50),         # short base: no sampled threshold
    (200_000, 4, 4, 21, 100),      # m = 4: the pre-filter of round 6
]


@pytest.mark.parametrize("n,m,sub,nq,K", SHAPES)
def test_non_finite_queries(rq, oracle, n, m, sub, nq, K):
    centers, queries, codes = _case(41, n, m, sub, nq)
    bad_nan, bad_pinf, bad_ninf = 3, 9, 12          # three different positions of two groups (8 queries per group)
    queries[bad_nan, 1] = np.nan
    queries[bad_pinf, sub * m - 1] = np.inf
    queries[bad_ninf, 0] = -np.inf
    d1, i1 = rq.linscan_aqd_query(codes, centers, queries, K)
    clean = np.ones(nq, bool)
    clean[bad_nan] = False
    d0, i0 = oracle.linscan_aqd_query(codes, centers, queries[clean], K)
    # every query but the NaN one -- the +-Inf ones included: all their distances are +Inf, ties part by id -- equals the reference
    assert np.array_equal(i1[clean], i0) and _eq_bits(d1[clean], d0)
    assert np.all(np.isinf(d1[bad_pinf])) and np.array_equal(i1[bad_pinf], np.arange(K, dtype=np.uint32))
    # the NaN query has no comparable distance at all: K times "no row"
    assert np.all(np.isnan(d1[bad_nan])) and np.all(i1[bad_nan] == 0xFFFFFFFF)


@pytest.mark.parametrize("n,m,sub,nq,K", SHAPES)
def test_non_finite_table_entries(rq, oracle, n, m, sub, nq, K):
    centers, queries, codes = _case(43, n, m, sub, nq)
    centers[2, 77, 1] = np.nan           # rows with code 77 in position 2: NaN distance to every query
    centers[m - 3, 3, 0] = np.inf        # rows with code 3 in position m - 3: +Inf distance to every query
    centers[0, 200, sub - 1] = -np.inf
    d1, i1 = rq.linscan_aqd_query(codes, centers, queries, K)
    keep = np.flatnonzero(codes[:, 2] != 77)
    d0, j0 = oracle.linscan_aqd_query(np.ascontiguousarray(codes[keep]), centers, queries, K)
    assert np.array_equal(i1, keep[j0].astype(np.uint32)) and _eq_bits(d1, d0)


def test_a_base_of_nan_rows_only(rq):
    """Fewer comparable rows than k: the comparable ones first, then (NaN, no row)."""
    centers, queries, codes = _case(47, 50_000, 8, 4, 10)
    centers[4, :, 0] = np.nan
    centers[4, 9, 0] = 0.25              # only rows with code 9 in position 4 have a distance
    K = 500
    d1, i1 = rq.linscan_aqd_query(codes, centers, queries, K)
    good = np.flatnonzero(codes[:, 4] == 9)
    assert 0 < good.size < K
    for q in range(queries.shape[0]):
        diff = centers - queries[q].reshape(8, 1, 4)
        lut = diff * diff
        acc = np.zeros(good.size, np.float32)
        for k in range(8):
            t = np.zeros(256, np.float32)
            for s in range(4):
                t = t + lut[k, :, s]
            acc = acc + t[codes[good, k]]
        order = np.lexsort((good, acc))
        assert np.array_equal(i1[q, :good.size], good[order].astype(np.uint32))
        assert _eq_bits(d1[q, :good.size], acc[order])
        assert np.all(np.isnan(d1[q, good.size:])) and np.all(i1[q, good.size:] == 0xFFFFFFFF)


def test_non_finite_inputs_through_linscan_pq_one_based(rq):
    """The Julia-side view (src/Linscan.jl:25: idx .+ 1): "no row" reads 0."""
    centers, queries, codes = _case(53, 20_000, 8, 4, 8)
    queries[5, 7] = np.nan
    C = [centers[i] for i in range(8)]
    d, idx = rq.linscan_pq(codes, queries, C, 64, 10)
    assert np.all(idx[5] == 0) and np.all(np.isnan(d[5]))
    assert np.all(idx[np.arange(8) != 5] >= 1)
