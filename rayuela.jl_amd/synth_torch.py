"""The generators of synth.py (integer-only, counter-based splitmix64) as torch int64
tensor programs, so bench.py can build the SIFT1M / Deep1M-shape base ON THE DEVICE in milliseconds and still
get exactly the bytes synth.sift_like / synth.deep_like produce on any box (tests/test_synth_torch.py compares
them).  Harness code: nothing here is on the product data path.

torch has no uint64 arithmetic; splitmix64 runs on int64 with wrap-around multiply/add (two's complement) and
logical shifts spelled (x >> s) & mask; unsigned `% p` is folded from the top 63 bits and the low bit."""
import torch

_C1 = -7046029254386353131          # 0x9E3779B97F4A7C15 as int64
_C2 = -4658895280553007687          # 0xBF58476D1CE4E5B9
_C3 = -7723592293110705685          # 0x94D049BB133111EB


def _lsr(x, s):
    return (x >> s) & ((1 << (64 - s)) - 1)


def splitmix64(x):
    z = x + _C1
    z = (z ^ _lsr(z, 30)) * _C2
    z = (z ^ _lsr(z, 27)) * _C3
    return z ^ _lsr(z, 31)


def _umod(z, p):
    """(z as uint64) % p for 0 < p < 2^31."""
    if p & (p - 1) == 0:
        return z & (p - 1)
    hi = _lsr(z, 1) % p
    return (hi * 2 + (z & 1)) % p


def _s64(v):
    v &= 0xFFFFFFFFFFFFFFFF
    return v - (1 << 64) if v >= (1 << 63) else v


def sift_like(n, d=128, seed=1234, ncentres=1024, row0=0, device="cuda", chunk=1 << 18):
    dev = torch.device(device)
    i64 = dict(dtype=torch.int64, device=dev)
    rows = torch.arange(row0, row0 + n, **i64)
    cid = _umod(splitmix64(rows ^ _s64(seed << 20)), ncentres)
    ce = torch.arange(ncentres * d, **i64)
    cent = _umod(splitmix64(ce ^ _s64(seed * 0x51ED27) ^ 0xC0FFEE), 128).to(torch.int32).view(ncentres, d)
    out = torch.empty((n, d), dtype=torch.float32, device=dev)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        e = torch.arange((row0 + a) * d, (row0 + b) * d, **i64)
        noise = torch.zeros(((b - a) * d,), dtype=torch.int32, device=dev)
        for t in range(4):
            hsh = splitmix64((e * 4 + t) ^ _s64(seed << 32))
            noise += _umod(hsh, 33).to(torch.int32) - 16
        v = cent[cid[a:b]] + noise.view(b - a, d)
        out[a:b] = v.clamp_(0, 255).to(torch.float32)
    return out


def _pairwise_sum_cols(sq, lo, n):
    """numpy's pairwise summation (the order np.sum uses along a contiguous axis) of columns [lo, lo+n) of sq."""
    if n < 8:
        res = torch.zeros_like(sq[:, 0]) if n == 0 else sq[:, lo].clone()
        for i in range(1, n):
            res = res + sq[:, lo + i]
        return res
    if n <= 128:
        r = [sq[:, lo + j] for j in range(8)]
        i = 8
        while i < n - (n % 8):
            r = [r[j] + sq[:, lo + i + j] for j in range(8)]
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < n:
            res = res + sq[:, lo + i]
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return _pairwise_sum_cols(sq, lo, n2) + _pairwise_sum_cols(sq, lo + n2, n - n2)


def deep_like(n, d=96, seed=1234, row0=0, device="cuda", chunk=1 << 18):
    dev = torch.device(device)
    out = torch.empty((n, d), dtype=torch.float32, device=dev)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        e = torch.arange((row0 + a) * d, (row0 + b) * d, dtype=torch.int64, device=dev)
        acc = torch.zeros(((b - a) * d,), dtype=torch.float64, device=dev)
        for t in range(4):
            hsh = splitmix64((e * 4 + t) ^ _s64(seed << 32) ^ 0xDEE9)
            acc += _lsr(hsh, 40).to(torch.float64) / float(1 << 24) - 0.5
        v = acc.view(b - a, d)
        nrm = torch.sqrt(_pairwise_sum_cols(v * v, 0, d)).clamp_min(1e-30)
        out[a:b] = (v / nrm[:, None]).to(torch.float32)
    return out


def random_codes(n, m, seed=1234, row0=0, device="cuda"):
    e = torch.arange(row0 * m, (row0 + n) * m, dtype=torch.int64, device=torch.device(device))
    return _lsr(splitmix64(e ^ _s64(seed)), 56).to(torch.uint8).view(n, m)
