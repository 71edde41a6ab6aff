"""GPU: randomized shapes (seeded, no hypothesis shrinking needed) of the scan and the encode against the
oracle -- ragged sizes, every supported m, tiny and huge k relative to n, duplicate rows, integer LUTs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _eq_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


@pytest.mark.parametrize("seed", range(24))
def test_scan_fuzz(rq, oracle, seed):
    import rayuela_jl_amd.synth as synth
    rng = np.random.default_rng(1000 + seed)
    m = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 16, 24, 32, 40, 64]))
    sub = int(rng.choice([1, 2, 3, 4, 8, 16]))
    n = int(rng.choice([1, 2, 63, 64, 65, 1000, 4097, 20000, 140000, 270000]))
    nq = int(rng.choice([1, 3, 8, 9, 33]))
    K = int(min(n, rng.choice([1, 2, 7, 64, 100, 1000, 3000])))
    if rng.random() < 0.5:   # integer-valued tables and few distinct rows -> massive exact ties
        centers = rng.integers(0, 4, (m, 256, sub)).astype(np.float32)
        queries = rng.integers(0, 4, (nq, m * sub)).astype(np.float32)
        codes = (synth.random_codes(n, m, seed=seed) % 3).astype(np.uint8)
    else:
        centers = rng.standard_normal((m, 256, sub)).astype(np.float32)
        queries = rng.standard_normal((nq, m * sub)).astype(np.float32)
        codes = synth.random_codes(n, m, seed=seed)
    d0, i0 = oracle.linscan_aqd_query(codes, centers, queries, K)
    d1, i1 = rq.linscan_aqd_query(codes, centers, queries, K)
    assert np.array_equal(i0, i1), (m, sub, n, nq, K)
    assert _eq_bits(d0, d1), (m, sub, n, nq, K)


@pytest.mark.parametrize("style", range(7))
@pytest.mark.parametrize("m,K", [(8, 10), (8, 1000), (16, 100), (4, 10), (4, 1000)])
def test_scan_prefilter_on_hostile_tables(rq, oracle, style, m, K):
    """Shapes that run the integer pre-filter (m in {4, 8, 16}, rows >= 64 k) on tables built to break a lower-bound
    filter: no contrast, one dominating sub-quantizer, massive ties, few distinct rows, a large common offset
    (tau ~ sum of the table minima), high contrast.  The filter may only ever cost time: ids and distance bits
    must equal the reference's."""
    import rayuela_jl_amd.synth as synth
    rng = np.random.default_rng(7000 + 10 * style + m)
    sub, n, nq = 2, 300_000, 11
    centers = rng.standard_normal((m, 256, sub)).astype(np.float32)
    queries = rng.standard_normal((nq, m * sub)).astype(np.float32)
    codes = synth.random_codes(n, m, seed=style)
    if style == 1:
        centers[:] = centers[:, :1, :]
    elif style == 2:
        centers[0] *= 1000.0
    elif style == 3:
        centers = rng.integers(0, 3, (m, 256, sub)).astype(np.float32)
        queries = rng.integers(0, 3, (nq, m * sub)).astype(np.float32)
    elif style == 4:
        codes = codes[rng.integers(0, 50, n)]
    elif style == 5:
        queries += 1000.0
    elif style == 6:
        centers[:, 8:, :] += 30.0
        queries *= 0.1
    d0, i0 = oracle.linscan_aqd_query(codes, centers, queries, K)
    d1, i1 = rq.linscan_aqd_query(codes, centers, queries, K)
    assert np.array_equal(i0, i1), (style, m, K)
    assert _eq_bits(d0, d1), (style, m, K)
    # the same answer with the filter switched off (tuning knob): it is an accelerator, not part of the result
    rq.set_tuning("SCAN_FILTER", 0)
    try:
        d2, i2 = rq.linscan_aqd_query(codes, centers, queries, K)
    finally:
        rq.set_tuning("SCAN_FILTER", 1)
    assert np.array_equal(i1, i2) and _eq_bits(d1, d2)
    from rayuela_jl_amd import _lib
    assert (_lib.lib().rq_last_scan_kernel() or b"").decode().startswith("adc_scan_kernel<%d, false, false" % m)      # the knob did switch it off
    if m == 8:
        # ... and with the other byte-table variant (6-bit entries, two sum sets: the library's choice for k >= 8192)
        rq.set_tuning("SCAN_FINE_MIN_K", 1)
        try:
            d3, i3 = rq.linscan_aqd_query(codes, centers, queries, K)
        finally:
            rq.set_tuning("SCAN_FINE_MIN_K", 0)
        assert np.array_equal(i1, i3) and _eq_bits(d1, d3)


@pytest.mark.parametrize("style", range(7))
@pytest.mark.parametrize("m,K", [(8, 10), (8, 1000), (16, 100)])
def test_lsq_prefilter_on_hostile_tables(rq, oracle, style, m, K):
    """linscan_lsq at sizes that run the LSQ pre-filter (signed tables -2<q,c>, row norms as a ninth byte table after
    |c|^2 has been folded into the tables): norms that are the true |x_hat|^2, constant, negative, with a huge spread,
    unrelated to the codebooks; near-orthogonal and heavily correlated codebooks; a few distinct rows (massive ties).
    Ids and distance bits must equal the oracle's (which is pinned against the compiled reference on the goldens),
    with the filter on and off."""
    import rayuela_jl_amd.synth as synth
    rng = np.random.default_rng(9000 + 10 * style + m)
    d, n, nq, h = 24, 200_000, 9, 256
    cb = rng.standard_normal((m * h, d)).astype(np.float32)
    queries = rng.standard_normal((nq, d)).astype(np.float32)
    codes = synth.random_codes(n, m, seed=100 + style)
    if style == 1:                       # near-orthogonal codebooks (PQ written as an additive quantizer)
        sub = d // m if d % m == 0 else 1
        cb[:] = 0
        for i in range(m):
            cb[i * h:(i + 1) * h, (i * sub) % d:(i * sub) % d + sub] = rng.standard_normal((h, sub)).astype(np.float32) * 3
    elif style == 2:                     # heavily correlated codebooks: large cross terms
        cb += rng.standard_normal((1, d)).astype(np.float32) * 5
    elif style == 5:
        codes = codes[rng.integers(0, 40, n)]
    xhat = np.zeros((n, d), dtype=np.float64)
    for i in range(m):
        xhat += cb[i * h + codes[:, i].astype(np.int64)]
    norms = (xhat ** 2).sum(1).astype(np.float32)
    if style == 3:
        norms[:] = 7.25                  # constant
    elif style == 4:
        norms = (rng.standard_normal(n) * 1e4).astype(np.float32)     # signed, unrelated, huge spread
    elif style == 6:
        norms = -norms
    d0, i0 = oracle.linscan_lsq(codes, cb, queries, norms, K)
    C = [cb[i * h:(i + 1) * h] for i in range(m)]
    R = np.eye(d, dtype=np.float32)
    d1, i1 = rq.linscan_lsq(codes, queries, C, norms, R, K)
    assert np.array_equal(i0.astype(np.int64), i1.astype(np.int64)), (style, m, K)
    assert _eq_bits(d0, d1), (style, m, K)
    rq.set_tuning("SCAN_FILTER_LSQ", 0)
    try:
        d2, i2 = rq.linscan_lsq(codes, queries, C, norms, R, K)
    finally:
        rq.set_tuning("SCAN_FILTER_LSQ", 1)
    assert np.array_equal(i1, i2) and _eq_bits(d1, d2)


@pytest.mark.parametrize("m,K", [(7, 100), (12, 50), (5, 10), (3, 10), (16, 1), (8, 1)])
def test_lsq_prefilter_padded_widths(rq, oracle, m, K):
    """m that is padded to the next tiled width (7 -> 8, 12 -> 16: the padding tables and their |c|^2 are zero) and widths
    without a pre-filter (3, 5), at sizes where the filter runs."""
    import rayuela_jl_amd.synth as synth
    rng = np.random.default_rng(m)
    d, n, nq, h = 20, 150_000, 7, 256
    cb = np.zeros((m * h, d), np.float32)
    for i in range(m):
        cb[i * h:(i + 1) * h, (2 * i) % d:(2 * i) % d + 2] = rng.standard_normal((h, 2)).astype(np.float32) * 3
    q = rng.standard_normal((nq, d)).astype(np.float32)
    codes = synth.random_codes(n, m, seed=m)
    xh = np.zeros((n, d))
    for i in range(m):
        xh += cb[i * h + codes[:, i].astype(np.int64)]
    norms = (xh ** 2).sum(1).astype(np.float32)
    d0, i0 = oracle.linscan_lsq(codes, cb, q, norms, K)
    d1, i1 = rq.linscan_lsq(codes, q, [cb[i * h:(i + 1) * h] for i in range(m)], norms, np.eye(d, dtype=np.float32), K)
    assert np.array_equal(i0.astype(np.int64), i1.astype(np.int64)) and _eq_bits(d0, d1)


@pytest.mark.parametrize("seed", range(16))
def test_encode_fuzz(rq, oracle, seed):
    import rayuela_jl_amd.synth as synth
    rng = np.random.default_rng(2000 + seed)
    m = int(rng.choice([1, 2, 3, 4, 7, 8, 16, 32]))
    sub = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 12, 16]))
    extra = int(rng.choice([0, 0, 1, m - 1])) if m > 1 else 0      # uneven splitarray splits
    d = m * sub + extra
    h = int(rng.choice([1, 2, 17, 32, 64, 100, 255, 256]))
    n = int(rng.choice([1, 31, 32, 33, 1000, 5003]))
    X = (rng.standard_normal((n, d)) * 5).astype(np.float32)
    if rng.random() < 0.3:
        X = np.round(X)                                            # exact ties between centroids happen
    off = synth.splitarray(d, m)
    C = [np.round(rng.standard_normal((h, int(off[i + 1] - off[i]))) * 5).astype(np.float32) for i in range(m)]
    if h > 2:
        C[0][h // 2] = C[0][0]                                     # duplicate centroid -> first index must win
    codes0 = oracle.encode_pq(X, synth.cat_codebooks(C), m, h)
    codes1 = rq.quantize_pq_u8(X, C)
    assert np.array_equal(codes0, codes1), (m, d, h, n, int((codes0 != codes1).sum()))
