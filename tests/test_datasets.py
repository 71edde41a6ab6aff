"""CPU: read_dataset / load_experiment_data (src/read_datasets.jl, demos/experiment_utils.jl:62-90) on a miniature
data directory written with the package's own xvecs writers and libhdf5 binding."""
import os

import numpy as np
import pytest


def test_load_experiment_data_sift_layout(tmp_path, rq):
    from rayuela_jl_amd import datasets
    rng = np.random.default_rng(0)
    d, nt, nb, nq, kgt = 8, 30, 50, 7, 5
    root = tmp_path / "data"
    os.makedirs(root / "sift")
    Xt = rng.random((nt, d)).astype(np.float32)
    Xb = rng.random((nb, d)).astype(np.float32)
    Xq = rng.random((nq + 3, d)).astype(np.float32)
    gt0 = rng.integers(0, nb, (nq + 3, kgt)).astype(np.int32)          # zero-based, k neighbours per query
    rq.fvecs_write(Xt, str(root / "sift" / "sift_learn.fvecs"))
    rq.fvecs_write(Xb, str(root / "sift" / "sift_base.fvecs"))
    rq.fvecs_write(Xq, str(root / "sift" / "sift_query.fvecs"))
    rq.ivecs_write(gt0, str(root / "sift" / "sift_groundtruth.ivecs"))
    a, b, q, gt = datasets.load_experiment_data("SIFT1M", 20, 40, nq, data_root=str(root))
    assert np.array_equal(a, Xt[:20]) and np.array_equal(b, Xb[:40]) and np.array_equal(q, Xq[:nq])
    assert gt.dtype == np.uint32 and np.array_equal(gt, gt0[:nq, 0].astype(np.uint32) + 1)   # one-based top neighbour
    assert np.array_equal(datasets.read_dataset("SIFT1M_base", (11, 15), data_root=str(root)), Xb[10:15])
    with pytest.raises(KeyError):
        datasets.read_dataset("nope", 3)


def test_hdf5_backed_dataset(tmp_path, rq):
    from rayuela_jl_amd import datasets, h5results
    if not h5results.available():
        pytest.skip("libhdf5 not found")
    rng = np.random.default_rng(1)
    root = tmp_path / "data"
    os.makedirs(root / "deep")
    f = str(root / "deep" / "deep.h5")
    base = rng.random((40, 6)).astype(np.float32)
    train = rng.random((25, 6)).astype(np.float32)
    query = rng.random((9, 6)).astype(np.float32)
    gt = rng.integers(1, 41, 9).astype(np.int64)                           # Deep1M's gt is one id per query already
    for name, arr in (("base", base), ("train", train), ("query", query), ("gt", gt)):
        h5results.h5write(f, name, arr)
    Xt, Xb, Xq, g = datasets.load_experiment_data("Deep1M", 20, 40, 5, data_root=str(root))
    assert np.array_equal(Xt, train[:20]) and np.array_equal(Xb, base) and np.array_equal(Xq, query[:5])
    assert np.array_equal(g, gt[:5].astype(np.uint32))
    assert np.array_equal(datasets.read_dataset("Deep1M_base", (3, 7), data_root=str(root)), base[2:7])
    with pytest.raises(ValueError):
        datasets.read_dataset("Deep1M", 26, data_root=str(root))        # the file holds 25
