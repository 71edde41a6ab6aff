"""CPU: the canonical encode oracle (k-ordered fmaf chains, oracle/rq_oracle.c) against the same published algorithm
evaluated with a real OpenBLAS sgemm / sdot (oracle/blas_order.py).  The reference's own summation order lives in its
OpenBLAS build; whatever that order is, it can only move assignments between float64 near-ties.  Measured here: how
many codes a real BLAS order moves, and that each of them is such a near-tie."""
import numpy as np
import pytest

from conftest import golden
from oracle import blas_order, oracle


def _split(C_cat, d, m, h):
    off = oracle.splitarray(d, m)
    out, p = [], 0
    flat = np.asarray(C_cat, dtype=np.float32).reshape(-1)
    for i in range(m):
        sub = off[i + 1] - off[i]
        out.append(flat[p:p + h * sub].reshape(h, sub))
        p += h * sub
    return out, off


@pytest.mark.parametrize("name", ["encode_sift_mini", "encode_deep_mini", "encode_h100", "encode_uneven"])
def test_goldens_blas_order_vs_canonical(name):
    g = golden(name)
    X = g["X"]
    m, h = int(g["m"]), int(g["h"])
    C_list, off = _split(g["C"], X.shape[1], m, h)
    canon = oracle.encode_pq(X, g["C"], m, h)
    assert np.array_equal(canon, g["codes"])
    blas = blas_order.encode_pq(X, C_list, off)
    flips, outside, worst = blas_order.near_tie_report(X, C_list, off, canon, blas)
    print("\n%s: %d of %d codes differ between the fmaf-chain order and %s, %d outside the near-tie bound (worst %.3f)"
          % (name, flips, canon.size, blas_order.blas_version(), outside, worst))
    assert outside == 0
    assert flips <= max(2, canon.size // 2000)


def test_synthetic_100k_blas_order_vs_canonical():
    from rayuela_jl_amd import synth
    n, d, m, h = 100_000, 128, 8, 256
    X = synth.sift_like(n, d, seed=synth.SEED_BASE)
    S = synth.sift_like(20_000, d, seed=synth.SEED_BASE, row0=3_100_000_000)
    C = synth.codebooks(S, m, h, seed=synth.SEED_CODEBOOK, iters=3, sample=20000)
    off = synth.splitarray(d, m)
    canon = oracle.encode_pq(X, synth.cat_codebooks(C), m, h)
    blas = blas_order.encode_pq(X, C, off)
    flips, outside, worst = blas_order.near_tie_report(X, C, off, canon, blas)
    print("\nsift-like 1e5 x 128: %d of %d codes differ (%s), %d outside the bound, worst %.3f"
          % (flips, canon.size, blas_order.blas_version(), outside, worst))
    assert outside == 0 and flips <= canon.size // 2000


def test_deep_like_opq_50k_blas_order_vs_canonical():
    """Non-integer data (unit-norm, rotated, m=16, sub=6): the minimum distances themselves differ in the last bit
    for a good part of the rows (sdot sums lane-partials, the chain does not) -- the assignments still agree."""
    from rayuela_jl_amd import synth
    n, d, m, h = 50_000, 96, 16, 256
    X = oracle.rotate_T(synth.rotation(d), synth.deep_like(n, d, seed=synth.SEED_BASE))
    C = synth.codebooks(X[:20000], m, h, seed=synth.SEED_CODEBOOK, iters=3, sample=20000)
    off = synth.splitarray(d, m)
    canon, costs = oracle.encode_pq(X, synth.cat_codebooks(C), m, h, with_costs=True)
    blas = blas_order.encode_pq(X, C, off)
    same_cost = 0
    for i in range(m):
        mn = blas_order.pairwise_sqeuclidean(C[i], X[:, off[i]:off[i + 1]]).min(1)
        same_cost += int((mn.view(np.uint32) == costs[:, i].view(np.uint32)).sum())
    flips, outside, worst = blas_order.near_tie_report(X, C, off, canon, blas)
    print("\ndeep-like 5e4 x 96 (m=16): %d of %d codes differ (%s); %.1f%% of the minimum distances are bit-equal; "
          "%d outside the bound" % (flips, canon.size, blas_order.blas_version(), 100.0 * same_cost / canon.size, outside))
    assert outside == 0 and flips <= canon.size // 2000


def test_rotation_blas_order_close_to_canonical():
    from rayuela_jl_amd import synth
    d = 96
    X = synth.deep_like(5000, d, seed=synth.SEED_BASE)
    R = synth.rotation(d)
    a = oracle.rotate_T(R, X)
    b = blas_order.rotate_T(R, X)
    ref = X.astype(np.float64) @ R.astype(np.float64).T      # RX[j][i] = sum_k Rc[i][k] X[j][k]
    assert np.abs(a - ref).max() < 2e-6 and np.abs(b - ref).max() < 2e-6
    print("\nrotation d=96: %.1f%% of the 4.8e5 outputs bit-equal between the fmaf chain and sgemm"
          % (100.0 * np.mean(a.view(np.uint32) == b.view(np.uint32))))
