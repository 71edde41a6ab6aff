"""GPU: page-locked result arrays (rq_host_alloc / rq_host_free) behind linscan_* and Index.search -- same answers,
buffers come from and go back to the library's pool, limits fall back to ordinary arrays."""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _eq_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def test_large_results_are_pinned_and_correct(rq, oracle):
    from rayuela_jl_amd import _lib, synth
    rng = np.random.default_rng(5)
    n, m, sub, nq, k = 20_000, 8, 4, 1200, 1000
    centers = rng.standard_normal((m, 256, sub)).astype(np.float32)
    queries = rng.standard_normal((nq, m * sub)).astype(np.float32)
    codes = synth.random_codes(n, m, seed=3)
    C = [centers[i] for i in range(m)]
    d0, i0 = oracle.linscan_aqd_query(codes, centers, queries[:64], k)
    dists, idx = rq.linscan_pq(codes, queries, C, 8 * m, k)
    assert isinstance(dists.base, _lib._PinnedBlock) and isinstance(idx.base, _lib._PinnedBlock)
    assert dists.flags.c_contiguous and dists.dtype == np.float32 and idx.dtype == np.uint32
    assert np.array_equal(idx[:64] - 1, i0) and _eq_bits(dists[:64], d0)
    # ordinary numpy semantics: slices keep the block alive, copies do not
    keep = dists[5:7]
    ptr = dists.base._ptr
    ref = dists[5:7].copy()
    del dists
    gc.collect()
    assert np.array_equal(keep, ref)
    # with the pool switched off the answer is the same, in numpy-owned memory
    rq.set_tuning("HOST_PIN", 0)
    try:
        d2, i2 = rq.linscan_pq(codes, queries, C, 8 * m, k)
    finally:
        rq.set_tuning("HOST_PIN", 1)
    assert not isinstance(d2.base, _lib._PinnedBlock)
    assert np.array_equal(i2, idx) and _eq_bits(d2[5:7], ref)
    del keep
    gc.collect()
    # the freed block is reused for the next result of that size
    d3, i3 = rq.linscan_pq(codes, queries, C, 8 * m, k)
    assert ptr in (d3.base._ptr, i3.base._ptr)
    assert np.array_equal(i3, idx)


def test_pool_limit_falls_back_to_plain_arrays(rq):
    from rayuela_jl_amd import _lib
    rq.set_tuning("HOST_PIN_MAX_MB", 16)
    try:
        a = _lib.result_empty((2 << 20,), np.float32)        # 8 MB
        b = _lib.result_empty((2 << 20,), np.float32)        # 8 MB: the limit is reached
        c = _lib.result_empty((2 << 20,), np.float32)
        assert isinstance(a.base, _lib._PinnedBlock) and isinstance(b.base, _lib._PinnedBlock)
        assert not isinstance(c.base, _lib._PinnedBlock)
        a[:] = 1.0
        b[:] = 2.0
        assert float(a.sum()) == float(2 << 20) and float(b[-1]) == 2.0
    finally:
        rq.set_tuning("HOST_PIN_MAX_MB", 0)
    del a, b, c
    gc.collect()
    assert _lib.lib().rq_release_workspaces() == 0     # also drops the idle page-locked buffers


def test_small_results_stay_in_numpy_memory(rq):
    from rayuela_jl_amd import _lib
    x = _lib.result_empty((100, 10), np.float32)
    assert x.flags.owndata


def test_large_results_take_the_chunked_copy_path(rq):
    """Results above HOST_DIRECT_MAX_MB are not stored over PCIe by the kernel but copied chunk by chunk behind the scans
    (k = 10000: 800 MB per 1e4 queries).  Forced here with a 1 MB limit; the answer is the same either way."""
    import rayuela_jl_amd.synth as synth
    rng = np.random.default_rng(8)
    m, sub, n, nq, K = 8, 4, 60_000, 9_000, 128           # 9000 queries >= 2 host chunks of 4096; 4.6 MB per result array: page-locked
    centers = rng.standard_normal((m, 256, sub)).astype(np.float32)
    C = [centers[i] for i in range(m)]
    Q = rng.standard_normal((nq, m * sub)).astype(np.float32)
    B = synth.random_codes(n, m, seed=3)
    d0, i0 = rq.linscan_pq(B, Q, C, 8 * m, K)
    rq.set_tuning("HOST_DIRECT_MAX_MB", 1)
    try:
        d1, i1 = rq.linscan_pq(B, Q, C, 8 * m, K)
        ts = rq.last_timing()
    finally:
        rq.set_tuning("HOST_DIRECT_MAX_MB", 0)
    assert np.array_equal(i0, i1) and np.array_equal(d0.view(np.uint32), d1.view(np.uint32))
    assert ts["d2h_ms"] > 0.0          # copies happened (the direct path reports 0)
