#!/usr/bin/env python3
"""LDS-pass model of the scan's byte-table gathers for a given row order (development aid).

A ds_read_b64 gather of a 32-lane group costs max over the 32 eight-byte slot columns of the number of DISTINCT
addresses in that column (same address = broadcast).  Group = rows {128w + 64h + 2j + r : j < 32} (lane j of half h,
sub-row r) for the m = 8 tiling, or 32 consecutive rows with --consecutive."""
import sys
import numpy as np


def passes(codes, consecutive=False):
    """mean passes per (group, sub-quantizer) -> array [m]"""
    n, m = codes.shape
    n32 = n // 128 * 128
    c = codes[:n32]
    if consecutive:
        g = c.reshape(-1, 32, m)
    else:
        g = c.reshape(-1, 2, 32, 2, m).transpose(0, 1, 3, 2, 4).reshape(-1, 32, m)   # [w][h][j][r] -> [w][h][r][j]
    G = g.shape[0]
    out = np.zeros(m)
    for k in range(m):
        b = g[:, :, k].astype(np.int64)                       # [G][32]
        # distinct addresses per slot: mark (group, value) presence
        pres = np.zeros((G, 256), dtype=bool)
        pres[np.arange(G)[:, None], b] = True
        load = pres.reshape(G, 8, 32).sum(1)                   # value = hi*32 + slot -> per slot count of distinct values
        out[k] = load.max(1).mean()
    return out


if __name__ == "__main__":
    codes = np.load(sys.argv[1])
    rng = np.random.default_rng(0)
    print("arrival ", passes(codes).round(2), passes(codes).sum().round(2))
    u = rng.integers(0, 256, codes.shape, dtype=np.uint8)
    print("uniform ", passes(u).round(2), passes(u).sum().round(2))
    key = np.zeros(len(codes), dtype=np.uint64)
    for k in range(codes.shape[1]):
        key = (key << np.uint64(8)) | codes[:, k].astype(np.uint64)
    o = np.argsort(key, kind="stable")
    p = passes(codes[o])
    print("lex     ", p.round(2), p.sum().round(2))


def bucket_order(codes, bits):
    """stable sort by the concatenated top bits[k] bits of byte k"""
    key = np.zeros(len(codes), dtype=np.uint64)
    for k, nb in enumerate(bits):
        if nb:
            key = (key << np.uint64(nb)) | (codes[:, k].astype(np.uint64) >> np.uint64(8 - nb))
    return np.argsort(key, kind="stable")


if __name__ == "__main__":
    for bits in ([3, 3, 3, 3, 3, 0, 0, 0], [2] * 8, [3, 3, 3, 3, 2, 1, 0, 0], [3, 3, 3, 3, 3, 3, 3, 3], [3, 3, 3, 3, 3, 1, 1, 1], [3,3,3,3,3,3,0,0]):
        o = bucket_order(codes, bits)
        for cons in (True, False):
            p = passes(codes[o], cons)
            print(bits, "consecutive" if cons else "stride2", p.round(2), p.sum().round(2))
