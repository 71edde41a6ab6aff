"""Full-size parity of BASELINE configs 3 and 4 on the whole-item scan paths (Next #2a).

Config 3: SIFT1M-shape OPQ (n = 1e6, d = 128, m = 8, rotation).  Config 4: Deep1M-shape OPQ (d = 96, m = 16 sub = 6): the
1024-thread m = 16 kernel.  nq = 4096 queries, so every query group is a WHOLE item (>= 512 groups at m = 8, >= 256 at
m = 16) exactly as in `bench.py --workload opq|deep`; k = 1000 (LDS cut + bitonic finish) and k = 10000 (the Julia
default, src/Linscan.jl:10: sample-sort finish and, at m = 8, the FINE 6-bit filter tables).  Data, codebooks and
rotation come from bench.py's generators.  Checked: the pinned oracle (the compiled reference when oracle/_ref is there)
on 64 sampled queries, bit for bit, and on ALL queries the size-independent properties -- ascending (dist, id) order,
unique in-range ids, and every returned distance recomputed on the device as the sequential-f32 ADC sum of that row."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _eq_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


_CACHE = {}


def _setup(wl):
    """The bench's inputs for workload `wl` ('opq' | 'deep'), encoded on the device: (codes, centers, rotated queries)."""
    if wl in _CACHE:
        return _CACHE[wl]
    import torch
    import rayuela_jl_amd.synth as synth
    import rayuela_jl_amd.synth_torch as st
    from rayuela_jl_amd import device as rqd
    dev = torch.device("cuda", 0)
    n, nq, h = 1_000_000, 4096, 256
    d, m = (96, 16) if wl == "deep" else (128, 8)

    def gen(rows, row0):
        if wl == "deep":
            return st.deep_like(rows, d, seed=synth.SEED_BASE, row0=row0, device=dev)
        return st.sift_like(rows, d, seed=synth.SEED_BASE, ncentres=65536, row0=row0, device=dev)
    Q = gen(nq, 3_000_000_000)
    S = gen(20_000, 3_100_000_000)
    R = torch.from_numpy(synth.rotation(d)).to(dev)
    C = synth.codebooks(rqd.rotate_T(R, S).cpu().numpy(), m, h, seed=synth.SEED_CODEBOOK, iters=3, sample=20000)
    Ccat = torch.from_numpy(synth.cat_codebooks(C)).to(dev)
    centers = torch.from_numpy(np.stack(C)).to(dev)
    X = torch.cat([gen(250_000, o) for o in range(0, n, 250_000)], 0)
    codes = rqd.encode_opq(X, R, Ccat, m, h)
    del X
    Qs = rqd.rotate_T(R, Q)                     # linscan_opq rotates the queries first (src/Linscan.jl:102)
    _CACHE.clear()                              # one workload's tensors at a time
    _CACHE[wl] = (codes, centers, Qs)
    return _CACHE[wl]


@pytest.mark.parametrize("K", [1000, 10000])
@pytest.mark.parametrize("wl", ["opq", "deep"])
def test_whole_item_scan_at_full_size(rq, oracle, wl, K):
    import torch
    from rayuela_jl_amd import device as rqd
    from rayuela_jl_amd import _lib
    codes, centers, Qs = _setup(wl)
    n, m = codes.shape
    nq = Qs.shape[0]
    plan = _lib.scan_plan(n, nq, m, Qs.shape[1], K)
    assert plan["whole"] == plan["groups"] and plan["slices"] == 1, plan      # the path bench.py takes
    dists, ids = rqd.linscan(codes, centers, Qs, K)
    ids64 = ids.long() & 0xFFFFFFFF
    # ascending lexicographic (dist, id); unique ids in range
    dd = dists[:, 1:] - dists[:, :-1]
    assert bool((dd >= 0).all())
    tie = dd == 0
    assert bool((ids64[:, 1:][tie] > ids64[:, :-1][tie]).all())
    assert bool((ids64 < n).all())
    srt = torch.sort(ids64, dim=1).values
    assert int((srt[:, 1:] == srt[:, :-1]).sum()) == 0
    # every reported distance is the sequential-f32 ADC distance of that row (deps/src/linscan_aqd.cpp:85-87)
    lut = rqd.adc_lut(centers, Qs)                               # [nq][m][256], bit-exact (tests/test_gpu_scan.py)
    for q0 in range(0, nq, 512):
        sl = slice(q0, q0 + 512)
        rows = codes[ids64[sl].reshape(-1)].long().reshape(-1, K, m)
        acc = torch.gather(lut[sl, 0, :], 1, rows[:, :, 0])
        for k in range(1, m):
            acc = acc + torch.gather(lut[sl, k, :], 1, rows[:, :, k])
        assert torch.equal(acc, dists[sl]), (wl, K, q0)
    # the pinned oracle on 64 queries spread over the batch (first and last group included)
    sel = np.unique(np.concatenate([np.arange(0, nq, nq // 62), [nq - 1, nq - 8]]))[:64]
    fn = oracle.ref_linscan_aqd_query if oracle.ref_available() else oracle.linscan_aqd_query
    d0, i0 = fn(codes.cpu().numpy(), centers.cpu().numpy(), Qs.cpu().numpy()[sel], K)
    assert np.array_equal(ids.cpu().numpy().view(np.uint32)[sel], i0), (wl, K)
    assert _eq_bits(dists.cpu().numpy()[sel], d0), (wl, K)
