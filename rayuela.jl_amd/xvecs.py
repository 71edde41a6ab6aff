"""Host mirror of src/xvecs_read.jl / src/xvecs_write.jl: TEXMEX .fvecs / .ivecs / .bvecs files.

Record layout (all little endian): int32 d, then d values (float32 / int32 / uint8).  Arrays are
returned in the memory-image convention of the other mirrors: (n, d) numpy == Julia's d-by-n matrix,
so a file read here can be handed to quantize_pq / linscan_pq as is.  SURVEY section 8f rank 4."""
import numpy as np


def _read(filename, bounds, dtype, itemsize):
    """bounds: None (all), n (first n) or (a, b) one-based inclusive like the reference's UnitRange."""
    with open(filename, "rb") as f:
        d = int(np.fromfile(f, dtype="<i4", count=1)[0])
        vecsizeof = 4 + d * itemsize
        f.seek(0, 2)
        vecnum = f.tell() // vecsizeof
        if bounds is None:
            a, b = 1, vecnum
        elif isinstance(bounds, (int, np.integer)):
            a, b = 1, int(bounds)
        else:
            a, b = int(bounds[0]), int(bounds[1])
        assert a >= 1                                   # xvecs_read.jl:18  @assert bounds.start >= 1
        n = b - a + 1
        f.seek((a - 1) * vecsizeof)
        raw = np.fromfile(f, dtype=np.uint8, count=vecsizeof * n)
    if raw.size != vecsizeof * n:
        raise EOFError("%s holds %d vectors, asked for %d..%d" % (filename, vecnum, a, b))
    raw = raw.reshape(n, vecsizeof)
    dims = raw[:, :4].copy().view("<i4").reshape(-1)
    assert (dims == d).all()                            # xvecs_read.jl:43-46: every record repeats d
    return np.ascontiguousarray(raw[:, 4:]).view(dtype).reshape(n, d)


def fvecs_read(bounds=None, filename=None):
    """fvecs_read(n_or_range, filename) -> (n, d) float32     (src/xvecs_read.jl:63-98)"""
    return _read(filename, bounds, "<f4", 4)


def ivecs_read(bounds=None, filename=None):
    """ivecs_read(n_or_range, filename) -> (n, d) int32       (src/xvecs_read.jl:109-144)"""
    return _read(filename, bounds, "<i4", 4)


def bvecs_read(bounds=None, filename=None):
    """bvecs_read(n_or_range, filename) -> (n, d) uint8       (src/xvecs_read.jl:14-52)"""
    return _read(filename, bounds, np.uint8, 1)


def _write(X, filename, dtype):
    X = np.ascontiguousarray(X, dtype=dtype)
    n, d = X.shape
    rec = np.empty((n, d + 1), dtype=dtype)
    rec[:, 0] = np.array([d], dtype="<i4").view(dtype)[0]   # reinterpret(Float32, Int32(d)), xvecs_write.jl:12
    rec[:, 1:] = X
    rec.tofile(filename)


def fvecs_write(X, filename):
    """fvecs_write(X, filename)    (src/xvecs_write.jl:10-16); X (n, d) float32."""
    _write(X, filename, "<f4")


def ivecs_write(X, filename):
    """ivecs_write(X, filename)    (src/xvecs_write.jl:19-25); X (n, d) int32."""
    _write(X, filename, "<i4")
