"""CPU: the LSQ / CQ scan restatement pinned against the outputs of the real
deps/src/linscan_aqd_pairwise_byte.cpp (golden .npz, and live when oracle/_ref is present)."""
import numpy as np
import pytest

from conftest import golden


@pytest.mark.parametrize("name", ["aq_mini_m8", "aq_mini_m4"])
def test_oracle_aq_matches_reference_golden(oracle, name):
    g = golden(name)
    for K in g["Ks"]:
        K = int(K)
        d, i = oracle.linscan_lsq(g["codes"], g["codebooks"], g["queries"], g["dbnorms"], K)
        assert np.array_equal(i, g["lsq_i%d" % K]) and np.array_equal(d.view(np.uint32), g["lsq_d%d" % K].view(np.uint32))
        d, i = oracle.linscan_cq(g["codes"], g["codebooks"], g["queries"], K)
        assert np.array_equal(i, g["cq_i%d" % K]) and np.array_equal(d.view(np.uint32), g["cq_d%d" % K].view(np.uint32))


def test_oracle_aq_vs_live_reference(oracle):
    if not oracle.ref_aq_available():
        pytest.skip("oracle/_ref/linscan_aqd_pairwise_byte.so not built (needs /root/reference)")
    import rayuela_jl_amd.synth as synth
    rng = np.random.default_rng(5)
    n, m, d, nq, K = 20000, 8, 24, 4, 500
    cb = rng.standard_normal((m * 256, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    codes = synth.random_codes(n, m, seed=9)
    nrm = rng.random(n).astype(np.float32)
    for fn, args in ((oracle.linscan_lsq, (codes, cb, q, nrm, K)), (oracle.linscan_cq, (codes, cb, q, K))):
        d0, i0 = fn(*args, use_ref=True)
        d1, i1 = fn(*args)
        assert np.array_equal(i0, i1) and np.array_equal(d0.view(np.uint32), d1.view(np.uint32))
