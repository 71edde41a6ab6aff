"""Encode parity against the REAL reference stack (Julia + Distances 0.8.0 + Clustering 0.12.2 + OpenBLAS).

The build image has no Julia, so the *_julia.i16 files do not exist here and the comparison tests SKIP (that is
the "parity unpinned" state DESIGN.md declares).  On any Julia box:  `julia julia/gen_encode_golden.jl`  writes
them next to the raw input mirrors in tests/golden/bin/; committing those outputs turns these tests on.
Every disagreement is classified with the float64 top-2 gap stored in the golden file: it must lie below the f32
rounding bound of the GEMM-trick distance (a near-tie, the one freedom the BLAS summation order has)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden

BIN = os.path.join(GOLDEN, "bin")
CASES = [ln.split() for ln in open(os.path.join(BIN, "cases.txt")).read().splitlines() if ln.strip()]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_raw_mirrors_equal_the_npz_inputs(case):
    name, kind, n, d, m, h, has_R = case[0], case[1], *map(int, case[2:])
    g = golden(name)
    X = np.fromfile(os.path.join(BIN, name + ".X.f32"), dtype="<f4").reshape(n, d)
    assert np.array_equal(X.view(np.uint32), g["X"].view(np.uint32))
    C = np.fromfile(os.path.join(BIN, name + ".C.f32"), dtype="<f4")
    assert np.array_equal(C.view(np.uint32), np.ascontiguousarray(g["C"]).reshape(-1).view(np.uint32))
    if has_R:
        R = np.fromfile(os.path.join(BIN, name + ".R.f32"), dtype="<f4").reshape(d, d)
        assert np.array_equal(R.view(np.uint32), g["R"].view(np.uint32))


def _scale(g, kind, m):
    """|x|^2 + max|c|^2 per (vector, sub-quantizer): what the f32 error of fl(fl(sa+sb) - 2g) is relative to."""
    X = g["X"].astype(np.float64)
    n, d = X.shape
    if kind == "rvq":
        C = g["C"].astype(np.float64)
        return np.stack([(X * X).sum(1) + (C[i] * C[i]).sum(1).max() for i in range(m)], axis=1)  # upper bound: |x| >= |residual|... loose on purpose
    per, extra = divmod(d, m)
    off = np.cumsum([0] + [per + (1 if i < extra else 0) for i in range(m)])
    Cf = g["C"].astype(np.float64)
    h = int(g["h"])
    out = np.zeros((n, m))
    pos = 0
    for i in range(m):
        w = off[i + 1] - off[i]
        Ci = Cf[pos:pos + w * h].reshape(h, w)
        pos += w * h
        Xs = X[:, off[i]:off[i + 1]]
        out[:, i] = (Xs * Xs).sum(1) + (Ci * Ci).sum(1).max()
    return out


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_julia_codes_vs_oracle(case):
    name, kind, n, d, m, h, has_R = case[0], case[1], *map(int, case[2:])
    path = os.path.join(BIN, name + ".codes_julia.i16")
    if not os.path.isfile(path):
        pytest.skip("no Julia output for %s (run julia/gen_encode_golden.jl on a Julia box): encode parity stays UNPINNED" % name)
    g = golden(name)
    Bj = np.fromfile(path, dtype="<i2").reshape(n, m).astype(np.int32) - 1     # Julia m x n column-major == [n][m]
    ours = g["codes"].astype(np.int32)
    diff = Bj != ours
    eps = 64 * 2.0 ** -24
    bad = diff & (g["gap64"] > eps * _scale(g, kind, m))
    print("%s: %d of %d codes differ from the Julia reference, %d outside the near-tie bound" % (name, int(diff.sum()), diff.size, int(bad.sum())))
    assert not bad.any()
    if has_R:
        po = os.path.join(BIN, name + ".codes_opq_julia.i16")
        if os.path.isfile(po):
            Bo = np.fromfile(po, dtype="<i2").reshape(n, m).astype(np.int32) - 1
            frac = float((Bo != g["codes_opq"].astype(np.int32)).mean())
            print("%s OPQ: %.4f%% of the codes differ (rotation order = OpenBLAS sgemm vs fmaf chain)" % (name, 100 * frac))
            assert frac < 2e-3
