"""CPU: the xvecs round trip of the reference's own test (test/xvecs.jl:3-19) on the host mirror."""
import numpy as np


def test_fvecs_ivecs_roundtrip(tmp_path, rq):
    rng = np.random.default_rng(0)
    d, n = 32, 1000
    X = (rng.random((n, d)) * 10).astype(np.float32)          # generate_random_dataset, test/common.jl:3-9
    fn = str(tmp_path / "x.fvecs")
    rq.fvecs_write(X, fn)
    assert np.array_equal(rq.fvecs_read(n, fn), X)
    assert np.array_equal(rq.fvecs_read(None, fn), X)
    assert np.array_equal(rq.fvecs_read((11, 20), fn), X[10:20])   # one-based inclusive range
    Xint = (np.floor(X - 0.5) * 1000).astype(np.int32)
    fn2 = str(tmp_path / "x.ivecs")
    rq.ivecs_write(Xint, fn2)
    assert np.array_equal(rq.ivecs_read(n, fn2), Xint)
    # record layout: int32 d followed by d values
    raw = np.fromfile(fn, dtype="<i4", count=1)
    assert raw[0] == d and (4 + 4 * d) * n == __import__("os").path.getsize(fn)


def test_bvecs_read(tmp_path, rq):
    rng = np.random.default_rng(1)
    d, n = 16, 50
    X = rng.integers(0, 256, (n, d)).astype(np.uint8)
    rec = np.zeros((n, 4 + d), dtype=np.uint8)
    rec[:, :4] = np.array([d], dtype="<i4").view(np.uint8)
    rec[:, 4:] = X
    fn = str(tmp_path / "x.bvecs")
    rec.tofile(fn)
    assert np.array_equal(rq.bvecs_read(n, fn), X)
    assert np.array_equal(rq.bvecs_read((5, 9), fn), X[4:9])
