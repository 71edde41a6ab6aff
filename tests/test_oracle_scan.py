"""CPU: pins oracle/rq_oracle.c (scan) against the reference's outputs.

The golden .npz files hold the outputs of the REAL reference scan (oracle/_ref, built from
/root/reference/deps/src/linscan_aqd.cpp -- see tests/gen_golden.py).  Where oracle/_ref is present
(build container, or shipped by gpurun) the restatement is also checked live against it,
including the reference's 1e7-row chunked path (deps/src/linscan_aqd.cpp:52-53,78-92).
"""
import numpy as np
import pytest

from conftest import golden

SCAN_CASES = ["scan_sift_mini", "scan_deep_mini", "scan_all_ties", "scan_dups", "scan_k_eq_n"]


@pytest.mark.parametrize("name", SCAN_CASES)
def test_oracle_scan_matches_reference_golden(oracle, name):
    g = golden(name)
    for K in g["Ks"]:
        dists, ids = oracle.linscan_aqd_query(g["codes"], g["centers"], g["queries"], int(K))
        assert np.array_equal(ids, g["ids_K%d" % K]), (name, K)
        assert np.array_equal(dists.view(np.uint32), g["dists_K%d" % K].view(np.uint32)), (name, K)


def test_all_ties_lowest_ids(oracle):
    g = golden("scan_all_ties")
    K = 17
    assert np.array_equal(g["ids_K%d" % K], np.tile(np.arange(K, dtype=np.uint32), (g["queries"].shape[0], 1)))


def test_oracle_lut_is_unfused_sequential(oracle):
    g = golden("scan_sift_mini")
    centers, q = g["centers"], g["queries"][0]
    lut = oracle.adc_lut(centers, q)
    m, h, sub = centers.shape
    exp = np.zeros((m, h), dtype=np.float32)
    for s in range(sub):
        diff = (centers[:, :, s] - q.reshape(m, sub)[:, s][:, None]).astype(np.float32)
        exp = (exp + (diff * diff).astype(np.float32)).astype(np.float32)
    assert np.array_equal(lut.view(np.uint32), exp.view(np.uint32))
    # distances: sequential f32 sum over sub-quantizers, checked on a few rows
    dist = oracle.adc_distances(g["codes"][:64], centers, q)
    acc = np.zeros(64, dtype=np.float32)
    for k in range(m):
        acc = (acc + lut[k, g["codes"][:64, k]]).astype(np.float32)
    assert np.array_equal(dist.view(np.uint32), acc.view(np.uint32))


@pytest.mark.parametrize("n,m,sub,nq,K", [(20000, 8, 16, 5, 1000), (3000, 16, 6, 4, 3000), (100, 4, 2, 7, 1),
                                           (1, 8, 4, 2, 1)])
def test_oracle_vs_live_reference(oracle, n, m, sub, nq, K):
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    import rayuela_jl_amd.synth as synth
    rng = np.random.default_rng(n + m)
    centers = rng.standard_normal((m, 256, sub)).astype(np.float32)
    queries = rng.standard_normal((nq, m * sub)).astype(np.float32)
    codes = synth.random_codes(n, m, seed=n)
    d0, i0 = oracle.ref_linscan_aqd_query(codes, centers, queries, K)
    d1, i1 = oracle.linscan_aqd_query(codes, centers, queries, K)
    assert np.array_equal(i0, i1)
    assert np.array_equal(d0.view(np.uint32), d1.view(np.uint32))


def test_oracle_vs_reference_chunked_path(oracle):
    """n > 1e7 + K exercises the reference's carried-top-K chunking; the restatement (global
    lexicographic top-K) must give the same answer, ties included."""
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    import rayuela_jl_amd.synth as synth
    n, m, sub, nq, K = 10_000_000 + 300_123, 2, 2, 2, 50
    centers = (synth.splitmix64(np.arange(m * 256 * sub, dtype=np.uint64)) % np.uint64(4)).astype(np.float32)
    centers = centers.reshape(m, 256, sub)
    queries = np.array([[0, 1, 2, 3], [3, 1, 0, 2]], dtype=np.float32)
    codes = synth.random_codes(n, m, seed=77)
    d0, i0 = oracle.ref_linscan_aqd_query(codes, centers, queries, K)
    d1, i1 = oracle.linscan_aqd_query(codes, centers, queries, K)
    assert np.array_equal(i0, i1)
    assert np.array_equal(d0.view(np.uint32), d1.view(np.uint32))
