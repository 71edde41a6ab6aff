"""CPU: the encode/rotation restatement (parity UNPINNED against the Julia reference -- see the
header of oracle/rq_oracle.c) against the committed canonical fixtures and an independent
float64 evaluation of the same formula (src/PQ.jl:40-41 call sites)."""
import numpy as np
import pytest

from conftest import golden

ENC = ["encode_sift_mini", "encode_deep_mini", "encode_uneven", "encode_h100"]


@pytest.mark.parametrize("name", ENC)
def test_encode_matches_golden_and_f64(oracle, name):
    g = golden(name)
    m, h = int(g["m"]), int(g["h"])
    X = g["X"]
    codes, costs = oracle.encode_pq(X, g["C"], m, h, with_costs=True)
    assert np.array_equal(codes, g["codes"])
    assert np.array_equal(costs.view(np.uint32), g["costs"].view(np.uint32))
    # any disagreement with float64 must be a near-tie (relative top-2 gap below f32 round-off)
    bad = codes != g["codes64"]
    if bad.any():
        scale = np.maximum(np.abs(costs[bad]), 1e-12)
        assert (g["gap64"][bad] / scale < 1e-4).all()
    assert bad.mean() < 1e-4


def test_splitarray_uneven(oracle):
    # src/utils.jl:179-203: first d%m parts carry one extra element
    assert oracle.splitarray(10, 4).tolist() == [0, 3, 6, 8, 10]
    assert oracle.splitarray(128, 8).tolist() == list(range(0, 129, 16))
    assert oracle.splitarray(96, 16).tolist() == list(range(0, 97, 6))
    assert oracle.splitarray(7, 7).tolist() == list(range(8))


def test_exact_hit_clamps_to_zero(oracle):
    g = golden("encode_sift_mini")
    # rows 0..63 were built to coincide with a centroid in subspace j % m -> cost exactly 0
    m = int(g["m"])
    for j in range(64):
        assert g["costs"][j, j % m] == 0.0


@pytest.mark.parametrize("name", ["encode_sift_mini", "encode_deep_mini"])
def test_rotation_and_opq(oracle, name):
    g = golden(name)
    RX = oracle.rotate_T(g["R"], g["X"])
    ref64 = g["X"].astype(np.float64) @ g["R"].astype(np.float64).T  # RX[j][i] = sum_k Rc[i][k] X[j][k]
    assert np.allclose(RX, ref64, rtol=1e-5, atol=1e-4)
    m, h = int(g["m"]), int(g["h"])
    assert np.array_equal(oracle.encode_opq(g["X"], g["R"], g["C"], m, h), g["codes_opq"])
    assert np.array_equal(oracle.encode_pq(RX, g["C"], m, h), g["codes_opq"])
    # R = I  ->  quantize_opq == quantize_pq (src/OPQ.jl:26)
    eye = np.eye(g["X"].shape[1], dtype=np.float32)
    assert np.array_equal(oracle.encode_opq(g["X"], eye, g["C"], m, h),
                          oracle.encode_pq(g["X"], g["C"], m, h))


def test_eval_recall_golden(oracle):
    g = golden("recall_tiny")
    r = oracle.eval_recall(g["gt"], g["idx"], int(g["k"]))
    assert np.allclose(r, g["recall"])


def test_eval_recall_searches_the_whole_column(oracle, rq):
    """src/Linscan.jl:207 looks for the ground-truth id in the WHOLE column of predictions, also when it is longer
    than k: a second occurrence beyond k makes the query a miss, a single occurrence beyond k counts for no R <= k."""
    k = 4
    gt = np.array([7, 8, 9, 5], dtype=np.uint32)
    idx = np.array([[7, 1, 2, 3, 4, 6],      # rank 1
                    [1, 8, 2, 3, 8, 6],      # twice (once beyond k) -> k+1
                    [1, 2, 3, 4, 6, 9],      # once, beyond k -> counts nowhere
                    [1, 2, 3, 5, 6, 0]], dtype=np.uint32)   # rank 4
    # by hand from :206-230: ranks = [1, 5, 6 -> excluded, 4]
    want = np.array([1, 1, 1, 2], dtype=np.float64) / 4.0
    assert np.allclose(oracle.eval_recall(gt, idx, k), want)
    assert np.allclose(rq.eval_recall(gt, idx, k, verbose=False), want)
