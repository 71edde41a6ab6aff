"""GPU parity of the additive-quantizer scans (linscan_lsq / linscan_cq): same scan
kernel, different LUT builder (+ per-row dbnorms).  Bar: ids and distances bit-exact with the outputs of
the reference's deps/src/linscan_aqd_pairwise_byte.cpp."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def _eq_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


@pytest.mark.parametrize("name", ["aq_mini_m8", "aq_mini_m4"])
def test_legacy_symbols_match_reference_golden(rq, name):
    g = golden(name)
    for K in g["Ks"]:
        K = int(K)
        d, i = rq.linscan_aqd_query_extra_byte(g["codes"], g["queries"], g["codebooks"], g["dbnorms"], K)
        assert np.array_equal(i, g["lsq_i%d" % K]) and _eq_bits(d, g["lsq_d%d" % K]), (name, K, "lsq")
        d, i = rq.linscan_aqd_query_extra_byte(g["codes"], g["queries"], g["codebooks"], None, K)
        assert np.array_equal(i, g["cq_i%d" % K]) and _eq_bits(d, g["cq_d%d" % K]), (name, K, "cq")


def test_linscan_lsq_and_cq_julia_conventions(rq, oracle):
    g = golden("aq_mini_m8")
    m = g["codes"].shape[1]
    d = g["queries"].shape[1]
    C = [g["codebooks"][k * 256:(k + 1) * 256] for k in range(m)]
    eye = np.eye(d, dtype=np.float32)
    dists, idx = rq.linscan_lsq(g["codes"].astype(np.int16) + 1, g["queries"], C, g["dbnorms"], eye, 100)
    assert np.array_equal(idx.view(np.int32), g["lsq_i100"]) and _eq_bits(dists, g["lsq_d100"])
    dists, idx = rq.linscan_cq(g["codes"], g["queries"], C, 100)
    assert np.array_equal(idx.view(np.int32), g["cq_i100"]) and _eq_bits(dists, g["cq_d100"])
    # a real rotation: linscan_lsq rotates the queries first (src/Linscan.jl:126)
    import rayuela_jl_amd.synth as synth
    R = synth.rotation(d, seed=3)
    d0, i0 = oracle.linscan_lsq(g["codes"], g["codebooks"], oracle.rotate_T(R, g["queries"]), g["dbnorms"], 50)
    d1, i1 = rq.linscan_lsq(g["codes"], g["queries"], C, g["dbnorms"], R, 50)
    assert np.array_equal(i1.view(np.int32), i0) and _eq_bits(d1, d0)


@pytest.mark.parametrize("n,m,d,nq,K", [(300_000, 8, 128, 20, 1000), (100_003, 16, 96, 9, 100), (5_000, 4, 32, 3, 5000)])
def test_aq_vs_oracle_random(rq, oracle, n, m, d, nq, K):
    import rayuela_jl_amd.synth as synth
    rng = np.random.default_rng(n + m)
    cb = rng.standard_normal((m * 256, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    codes = synth.random_codes(n, m, seed=n)
    nrm = (rng.random(n) * 50).astype(np.float32)
    d0, i0 = oracle.linscan_lsq(codes, cb, q, nrm, K)
    d1, i1 = rq.linscan_aqd_query_extra_byte(codes, q, cb, nrm, K)
    assert np.array_equal(i0, i1) and _eq_bits(d0, d1)
    d0, i0 = oracle.linscan_cq(codes, cb, q, K)
    d1, i1 = rq.linscan_aqd_query_extra_byte(codes, q, cb, None, K)
    assert np.array_equal(i0, i1) and _eq_bits(d0, d1)


@pytest.mark.parametrize("n,m,d,nq,K", [(300_000, 8, 64, 24, 1000), (100_003, 16, 48, 9, 100), (50_000, 5, 40, 7, 50), (5_000, 4, 32, 3, 5000)])
def test_prepared_lsq_index_equals_linscan_lsq(rq, oracle, n, m, d, nq, K):
    """rq_lsq_prepare / rq_lsq_search / rq_lsq_release: the pre-filter's O(n) preprocessing is paid once per base;
    every search returns what linscan_lsq returns (= the compiled reference's deps/src/linscan_aqd_pairwise_byte.cpp:14-94
    semantics, one-based ids), with and without a rotation, on clustered codes where the filter is active (m = 8, 16), on a
    padded width (m = 5) and on a width without the filter (m = 4)."""
    import rayuela_jl_amd.synth as synth
    rng = np.random.default_rng(n + m)
    C = [rng.standard_normal((256, d)).astype(np.float32) for _ in range(m)]
    cb = np.concatenate(C)
    codes = synth.random_codes(n, m, seed=n)
    codes[: n // 2] = codes[rng.integers(0, 300, n // 2)]            # half of the rows are copies: ties and a live filter
    nrm = ((cb.reshape(m, 256, d)[np.arange(m)[None, :], codes]).sum(1) ** 2).sum(1).astype(np.float32)   # true |x_hat|^2
    R = synth.rotation(d, seed=5)
    with rq.LsqIndex(codes, C, nrm) as ix:
        for rot in (None, R):
            for rep in range(2):                                      # searches re-use the prepared base
                q = rng.standard_normal((nq, d)).astype(np.float32)
                qr = q if rot is None else oracle.rotate_T(rot, q)
                d0, i0 = oracle.linscan_lsq(codes, cb, qr, nrm, K)
                d1, i1 = ix.search(q, rot, K)
                assert np.array_equal(i1.view(np.int32), i0) and _eq_bits(d1, d0), (m, rot is not None, rep)
                d2, i2 = rq.linscan_lsq(codes, q, C, nrm, np.eye(d, dtype=np.float32) if rot is None else rot, K)
                assert np.array_equal(i2, i1) and _eq_bits(d2, d1)
