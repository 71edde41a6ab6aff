"""CPU: the RVQ restatement (oracle/rq_oracle.c:oracle_encode_rvq, src/RVQ.jl:18-66) against a float64
brute-force restatement of the same loop and its own invariants.  Encode parity is UNPINNED against
Julia (see the oracle header): these tests pin the oracle's structure, not the BLAS summation order."""
import numpy as np
import pytest


def _rvq_problem(n, d, m, h, seed):
    rng = np.random.default_rng(seed)
    X = (rng.standard_normal((n, d)) * 20).astype(np.float32)
    C = np.empty((m, h, d), dtype=np.float32)
    Xr = X.copy()
    for i in range(m):   # crude residual codebooks: random residual points, shrinking scale
        C[i] = Xr[rng.choice(n, size=h, replace=n < h)] * np.float32(0.9)
        dist = ((Xr[:, None, :].astype(np.float64) - C[i][None].astype(np.float64)) ** 2).sum(-1)
        Xr = Xr - C[i][dist.argmin(1)]
    return X, C


@pytest.mark.parametrize("n,d,m,h", [(300, 32, 4, 16), (257, 24, 3, 33), (64, 128, 2, 256)])
def test_rvq_matches_float64_bruteforce(oracle, n, d, m, h):
    X, C = _rvq_problem(n, d, m, h, seed=n + d)
    codes, counts, Xr = oracle.encode_rvq(X, C, with_extras=True)
    assert codes.shape == (n, m) and codes.dtype == np.uint8
    # replay: the residual is the plain f32 subtraction of the chosen entries, stage by stage
    R = X.copy()
    for i in range(m):
        d2 = ((R[:, None, :].astype(np.float64) - C[i][None].astype(np.float64)) ** 2).sum(-1)
        best = d2.min(1)
        mine = d2[np.arange(n), codes[:, i]]
        # the f32 GEMM-trick distance may pick a different centre only on near-ties
        assert np.all(mine <= best * (1 + 1e-4) + 1e-2)
        R = R - C[i][codes[:, i]]
        assert np.array_equal(np.bincount(codes[:, i], minlength=h).astype(np.uint32), counts[i])
    assert np.array_equal(R.view(np.uint32), Xr.view(np.uint32))


def test_rvq_stage_equals_single_subquantizer_encode(oracle):
    X, C = _rvq_problem(200, 16, 3, 32, seed=5)
    codes = oracle.encode_rvq(X, C)
    first = oracle.encode_pq(X, C[0].reshape(-1), 1, 32)
    assert np.array_equal(codes[:, :1], first)
    # exact hit: a data point equal to a codebook entry of stage 1 costs max(.,0) = 0 there
    X2 = X.copy()
    X2[7] = C[0][3]
    assert oracle.encode_rvq(X2, C)[7, 0] == 3 or np.array_equal(C[0][oracle.encode_rvq(X2, C)[7, 0]], C[0][3])


@pytest.mark.parametrize("name", ["rvq_sift_mini", "rvq_deep_mini"])
def test_rvq_golden(oracle, name):
    from conftest import golden
    g = golden(name)
    codes, counts, Xr = oracle.encode_rvq(g["X"], g["C"], with_extras=True)
    assert np.array_equal(codes, g["codes"])
    assert np.array_equal(counts, g["counts"])
    assert np.array_equal(Xr.view(np.uint32), g["Xr"].view(np.uint32))
    # every f32-vs-f64 disagreement is a near-tie of the f64 distances
    flips = codes != g["codes64"]
    assert flips.mean() < 0.02
    assert codes[3, 0] == 5 or np.array_equal(g["C"][0][codes[3, 0]], g["C"][0][5])
