"""Deterministic synthetic data for the parity tests and bench.py.

Integer-only, counter-based (splitmix64) so the same arrays come out bit-identical on any box
and can be regenerated in C/HIP (csrc/rq_synth.hip generates the SIFT1B-shape codes on device
with the same hash).  Nothing here is on the product data path.

  sift_like(n, d, seed)   SIFT1M-shape vectors: 1024 centres with coords U{0..127}; each vector
                          = clip(centre + sum of 4 U{-16..16}, 0, 255) as f32 (integer valued).
  deep_like(n, d, seed)   Deep1M-shape: sum-of-4-uniform ~ Gaussian, L2-normalised in f64 -> f32.
  codebooks(X, m, h, seed, iters)  h sampled sub-vectors per subspace + a few Lloyd iterations
                          (harness-side, untimed; training is out of scope).
  rotation(d, seed)       orthonormal U V' from the f64 SVD of a seeded d x d matrix, as f32.
  random_codes(n, m, seed)  code[i][k] = splitmix64(seed ^ (i*m+k)) >> 56.
Seeds used everywhere: base 1234, queries 4321, codebooks 99, rotation 7.
"""
import numpy as np

SEED_BASE, SEED_QUERY, SEED_CODEBOOK, SEED_ROTATION = 1234, 4321, 99, 7
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _counter(n_elems, start=0):
    return np.arange(start, start + n_elems, dtype=np.uint64)


def sift_like(n, d=128, seed=SEED_BASE, ncentres=1024, row0=0):
    seed = np.uint64(seed)
    cid = (splitmix64(_counter(n, row0) ^ (seed << np.uint64(20))) % np.uint64(ncentres)).astype(np.int64)
    cent = (splitmix64(_counter(ncentres * d) ^ (seed * np.uint64(0x51ED27)) ^ np.uint64(0xC0FFEE))
            % np.uint64(128)).astype(np.int32).reshape(ncentres, d)
    out = np.empty((n, d), dtype=np.float32)
    step = max(1, (1 << 22) // d)
    for a in range(0, n, step):
        b = min(n, a + step)
        e = _counter((b - a) * d, (row0 + a) * d)
        noise = np.zeros((b - a) * d, dtype=np.int32)
        for t in range(4):
            hsh = splitmix64((e * np.uint64(4) + np.uint64(t)) ^ (seed << np.uint64(32)))
            noise += (hsh % np.uint64(33)).astype(np.int32) - 16
        v = cent[cid[a:b]] + noise.reshape(b - a, d)
        out[a:b] = np.clip(v, 0, 255).astype(np.float32)
    return out


def deep_like(n, d=96, seed=SEED_BASE, row0=0):
    seed = np.uint64(seed)
    out = np.empty((n, d), dtype=np.float32)
    step = max(1, (1 << 22) // d)
    for a in range(0, n, step):
        b = min(n, a + step)
        e = _counter((b - a) * d, (row0 + a) * d)
        acc = np.zeros((b - a) * d, dtype=np.float64)
        for t in range(4):
            hsh = splitmix64((e * np.uint64(4) + np.uint64(t)) ^ (seed << np.uint64(32)) ^ np.uint64(0xDEE9))
            acc += (hsh >> np.uint64(40)).astype(np.float64) / float(1 << 24) - 0.5
        v = acc.reshape(b - a, d)
        v /= np.maximum(np.sqrt((v * v).sum(axis=1, keepdims=True)), 1e-30)
        out[a:b] = v.astype(np.float32)
    return out


def splitarray(d, m):
    """src/utils.jl:179-203 as zero-based offsets (m+1 entries)."""
    per, extra = divmod(d, m)
    off = [0]
    for i in range(m):
        off.append(off[-1] + per + (1 if i < extra else 0))
    return np.asarray(off, dtype=np.int32)


def codebooks(X, m, h=256, seed=SEED_CODEBOOK, iters=5, sample=20000):
    """Returns a list of m arrays [h][sub_i] f32 (C view of Julia's sub x h matrices)."""
    n, d = X.shape
    off = splitarray(d, m)
    idx = (splitmix64(_counter(min(sample, n)) ^ np.uint64(seed * 7919)) % np.uint64(n)).astype(np.int64)
    S = X[idx].astype(np.float64)
    out = []
    for i in range(m):
        Xs = S[:, off[i]:off[i + 1]]
        pick = (splitmix64(_counter(h) ^ np.uint64(seed * 104729 + i)) % np.uint64(Xs.shape[0])).astype(np.int64)
        Ci = Xs[pick].copy()
        # de-duplicate identical seeds a little so Lloyd has something to do
        Ci += 1e-3 * ((splitmix64(_counter(Ci.size) ^ np.uint64(i + 17)) % np.uint64(1000)).astype(np.float64)
                      .reshape(Ci.shape) / 1000.0 - 0.5)
        for _ in range(iters):
            d2 = (Xs * Xs).sum(1)[:, None] - 2.0 * Xs @ Ci.T + (Ci * Ci).sum(1)[None, :]
            a = d2.argmin(1)
            for k in range(h):
                sel = a == k
                if sel.any():
                    Ci[k] = Xs[sel].mean(0)
        out.append(np.ascontiguousarray(Ci, dtype=np.float32))
    return out


def rvq_codebooks(X, m, h=256, seed=SEED_CODEBOOK, iters=3, sample=4096):
    """m residual codebooks [m][h][d] f32: stage i = a few Lloyd iterations on the residual of a sample
    (float64 brute force; a stand-in for train_rvq, src/RVQ.jl:86-127, good enough to give every stage
    a sensible scale)."""
    n, d = X.shape
    idx = (splitmix64(_counter(min(sample, n)) ^ np.uint64(seed * 7919)) % np.uint64(n)).astype(np.int64)
    Xr = X[idx].astype(np.float32).copy()
    out = np.empty((m, h, d), dtype=np.float32)
    for i in range(m):
        Ci = codebooks(Xr, 1, h, seed=seed + 1000 * (i + 1), iters=iters, sample=Xr.shape[0])[0]
        out[i] = Ci
        S = Xr.astype(np.float64)
        d2 = (S * S).sum(1)[:, None] - 2.0 * S @ Ci.T.astype(np.float64) + (Ci.astype(np.float64) ** 2).sum(1)[None, :]
        Xr = Xr - Ci[d2.argmin(1)]
    return out


def rotation(d, seed=SEED_ROTATION):
    A = (splitmix64(_counter(d * d) ^ np.uint64(seed * 31337)) >> np.uint64(11)).astype(np.float64)
    A = A / float(1 << 53) - 0.5
    U, _, Vt = np.linalg.svd(A.reshape(d, d))
    return np.ascontiguousarray(U @ Vt, dtype=np.float32)


def random_codes(n, m, seed=SEED_BASE, row0=0):
    e = _counter(n * m, row0 * m)
    return (splitmix64(e ^ np.uint64(seed)) >> np.uint64(56)).astype(np.uint8).reshape(n, m)


def cat_codebooks(C):
    """list of [h][sub_i] -> flat f32 concat (== cat(C...,dims=3) when all sub_i are equal)."""
    return np.concatenate([np.ascontiguousarray(c, dtype=np.float32).reshape(-1) for c in C])
