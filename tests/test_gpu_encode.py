"""GPU parity tests of the encode / rotation kernels vs the committed canonical fixtures and the
oracle.  Bar: codes bit-exact, rotated vectors bit-exact (the MFMA k-loop is the oracle's fmaf chain)."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu

ENC = ["encode_sift_mini", "encode_deep_mini", "encode_uneven", "encode_h100"]


def _split(Ccat, d, m, h):
    import rayuela_jl_amd.synth as synth
    off = synth.splitarray(d, m)
    out, pos = [], 0
    for i in range(m):
        sub = int(off[i + 1] - off[i])
        out.append(Ccat[pos:pos + h * sub].reshape(h, sub))
        pos += h * sub
    return out


@pytest.mark.parametrize("waves", [8, 16])
@pytest.mark.parametrize("name", ENC)
def test_quantize_pq_matches_golden(rq, name, waves):
    g = golden(name)
    m, h = int(g["m"]), int(g["h"])
    C = _split(g["C"], g["X"].shape[1], m, h)
    rq.set_tuning("ENC_WAVES", waves)
    try:
        B = rq.quantize_pq(g["X"], C)
    finally:
        rq.set_tuning("ENC_WAVES", 16)
    assert B.dtype == np.int16 and B.shape == g["codes"].shape
    assert np.array_equal(B, g["codes"].astype(np.int16) + 1)       # one-based (src/PQ.jl:45-47)
    assert np.array_equal(rq.quantize_pq_u8(g["X"], C), g["codes"])


@pytest.mark.parametrize("name", ["encode_sift_mini", "encode_deep_mini"])
def test_rotation_and_quantize_opq(rq, oracle, name):
    g = golden(name)
    m, h = int(g["m"]), int(g["h"])
    C = _split(g["C"], g["X"].shape[1], m, h)
    RX = rq.rotate(g["R"], g["X"])
    RX0 = oracle.rotate_T(g["R"], g["X"])
    assert np.array_equal(RX.view(np.uint32), RX0.view(np.uint32))
    B = rq.quantize_opq(g["X"], g["R"], C)
    assert np.array_equal(B, g["codes_opq"].astype(np.int16) + 1)
    eye = np.eye(g["X"].shape[1], dtype=np.float32)
    assert np.array_equal(rq.quantize_opq(g["X"], eye, C), rq.quantize_pq(g["X"], C))


@pytest.mark.parametrize("n,d,m,h,kind", [
    (100_017, 128, 8, 256, "sift"),   # ragged last tile
    (40_000, 96, 16, 256, "deep"),
    (5_000, 64, 4, 256, "deep"),      # sub = 16, m = 4
    (3_000, 50, 7, 33, "sift"),       # uneven split 8,7,7,7,7,7,7 and odd h
    (31, 16, 16, 17, "deep"),         # sub = 1, fewer rows than one tile
    (2_000, 256, 4, 256, "deep"),     # sub = 64
    (1_500, 960, 8, 256, "sift"),     # GIST shape: sub = 120 -> chunked wide kernel
    (700, 200, 1, 256, "deep"),       # one full-dimensional codebook (k-means assignment), 4 chunks of 32 k-steps... at NT=8: 7 of 16
    (900, 150, 2, 100, "sift"),       # sub = 75 (odd), h = 100
    (400, 131, 1, 33, "deep"),        # odd width, NT = 2
    (2_000, 128, 1, 256, "sift"),     # exactly 128 wide: registers-only KS = 64 kernel
    (2_000, 96, 1, 64, "deep"),       # exactly 96 wide: KS = 48
])
def test_encode_vs_oracle_random(rq, oracle, n, d, m, h, kind):
    import rayuela_jl_amd.synth as synth
    X = synth.sift_like(n, d, seed=n) if kind == "sift" else synth.deep_like(n, d, seed=n)
    C = synth.codebooks(X, m, h, seed=n + 1, iters=1, sample=min(n, 2000))
    codes0 = oracle.encode_pq(X, synth.cat_codebooks(C), m, h)
    codes1 = rq.quantize_pq_u8(X, C)
    assert np.array_equal(codes0, codes1), int((codes0 != codes1).sum())


def test_rotation_odd_dims(rq, oracle):
    import rayuela_jl_amd.synth as synth
    for d, n in [(10, 100), (33, 77), (96, 1000), (128, 4097)]:
        X = synth.deep_like(n, d, seed=d)
        R = synth.rotation(d, seed=d)
        assert np.array_equal(rq.rotate(R, X).view(np.uint32), oracle.rotate_T(R, X).view(np.uint32)), d


def test_rotation_and_opq_beyond_the_lds_resident_widths(rq, oracle):
    """d where R (d*d*4 B) no longer fits the LDS: the chunked rotation kernel keeps the oracle's k-ordered chain;
    quantize_opq at GIST-960 / MNIST-784 width (these returned RQ_EUNSUPPORTED)."""
    import rayuela_jl_amd.synth as synth
    for d, n in [(200, 333), (257, 100), (784, 600), (960, 1001)]:
        X = synth.deep_like(n, d, seed=d)
        R = synth.rotation(d, seed=d)
        assert np.array_equal(rq.rotate(R, X).view(np.uint32), oracle.rotate_T(R, X).view(np.uint32)), d
    d, n, m, h = 960, 3000, 8, 64
    X = synth.deep_like(n, d, seed=1)
    R = synth.rotation(d, seed=2)
    C = synth.codebooks(oracle.rotate_T(R, X), m, h, seed=3, iters=1, sample=2000)
    assert np.array_equal(rq.quantize_opq(X, R, C), oracle.encode_opq(X, R, synth.cat_codebooks(C), m, h).astype(np.int16) + 1)
    # train_opq at a width above 128 (update_centers in dimension chunks, rotation chunked)
    Xt = synth.deep_like(4000, 200, seed=4)
    Ct, Bt, Rt, obj = rq.train_opq(Xt, 4, 32, 3, "natural", seed=1)
    assert (np.diff(obj) <= 1e-5 * obj[:-1]).all() and np.abs(Rt @ Rt.T - np.eye(200)).max() < 1e-4
    assert np.array_equal(rq.quantize_opq(Xt, Rt, Ct), Bt)


def test_device_entry_points_and_full_size(rq, oracle):
    """SIFT1M-shape encode at full size on resident data; oracle-checked on a 60k-row sample,
    plus determinism (two runs identical)."""
    import torch
    import rayuela_jl_amd.synth as synth
    from rayuela_jl_amd import device as rqd
    n, d, m, h = 1_000_000, 128, 8, 256
    g = torch.Generator(device="cuda").manual_seed(5)
    X = torch.randint(0, 200, (n, d), generator=g, device="cuda").float()
    Xs = X[:20000].cpu().numpy()
    C = synth.codebooks(Xs, m, h, seed=3, iters=1, sample=4000)
    Ccat = torch.from_numpy(synth.cat_codebooks(C)).cuda()
    codes = rqd.encode_pq(X, Ccat, m, h)
    codes2 = rqd.encode_pq(X, Ccat, m, h)
    assert torch.equal(codes, codes2)
    sel = torch.arange(0, n, 17, device="cuda")[:60000]
    ref = oracle.encode_pq(X[sel].cpu().numpy(), synth.cat_codebooks(C), m, h)
    assert np.array_equal(codes[sel].cpu().numpy(), ref)
    # OPQ on resident data: rotate (MFMA) then encode == oracle on the sample
    R = torch.from_numpy(synth.rotation(d)).cuda()
    codes_o = rqd.encode_opq(X, R, Ccat, m, h)
    ref_o = oracle.encode_opq(X[sel].cpu().numpy(), R.cpu().numpy(), synth.cat_codebooks(C), m, h)
    assert np.array_equal(codes_o[sel].cpu().numpy(), ref_o)


def test_host_path_pipelined_chunks_and_device_list(rq, oracle, monkeypatch):
    """rq_encode_* uploads X in chunks while the previous chunk is encoded; with RAYUELA_HIP_DEVICES the rows are
    split over the listed devices (here the same GPU three times: three host threads, three row ranges)."""
    import rayuela_jl_amd.synth as synth
    n, d, m, h = 300_001, 128, 8, 256          # > one 262144-row chunk, ragged
    X = synth.sift_like(n, d, seed=77)
    C = synth.codebooks(X, m, h, seed=78, iters=1, sample=4000)
    R = synth.rotation(d, seed=79)
    ref = oracle.encode_pq(X, synth.cat_codebooks(C), m, h)
    ref_o = oracle.encode_opq(X, R, synth.cat_codebooks(C), m, h)
    assert np.array_equal(rq.quantize_pq_u8(X, C), ref)
    assert np.array_equal(rq.quantize_opq(X, R, C), ref_o.astype(np.int16) + 1)
    monkeypatch.setenv("RAYUELA_HIP_DEVICES", "0,0,0")
    assert np.array_equal(rq.quantize_pq_u8(X, C), ref)
    assert np.array_equal(rq.quantize_pq(X, C), ref.astype(np.int16) + 1)
    assert np.array_equal(rq.quantize_opq(X, R, C), ref_o.astype(np.int16) + 1)
    monkeypatch.delenv("RAYUELA_HIP_DEVICES")
    rq.set_tuning("HOST_OVERLAP", 0)
    try:
        assert np.array_equal(rq.quantize_pq_u8(X, C), ref)
    finally:
        rq.set_tuning("HOST_OVERLAP", 1)


def test_resident_dataset_encodes_match_the_host_calls(rq, oracle):
    import rayuela_jl_amd.synth as synth
    n, d, m, h = 50_000, 96, 16, 256
    X = synth.deep_like(n, d, seed=5)
    C = synth.codebooks(X, m, h, seed=6, iters=1, sample=4000)
    C2 = synth.codebooks(X, 8, 64, seed=7, iters=1, sample=4000)
    R = synth.rotation(d, seed=8)
    with rq.Dataset(X) as ds:
        assert np.array_equal(ds.quantize(C), rq.quantize_pq(X, C))
        assert np.array_equal(ds.quantize(C, R=R), rq.quantize_opq(X, R, C))
        assert np.array_equal(ds.quantize(C2, one_based=False), oracle.encode_pq(X, synth.cat_codebooks(C2), 8, 64))
        assert np.array_equal(ds.quantize(C), rq.quantize_pq(X, C))        # X itself is untouched by the OPQ call


# ---- encode_pq_split_kernel: bf16 matrix-core filter + exact re-evaluation of the candidates ----------------------------
def _enc_both(rq, oracle, X, C, m, h):
    """codes of the split kernel (default for even sub-space widths <= 16), of the f32-MFMA kernel and of the oracle"""
    Ccat = np.concatenate([np.ascontiguousarray(c, dtype=np.float32).reshape(-1) for c in C])
    ref = oracle.encode_pq(X, Ccat, m, h)
    got = rq.quantize_pq_u8(X, C)
    rq.set_tuning("ENC_SPLIT", 0)
    try:
        old = rq.quantize_pq_u8(X, C)
    finally:
        rq.set_tuning("ENC_SPLIT", 1)
    return got, old, ref


@pytest.mark.parametrize("sub", [2, 4, 6, 8, 10, 12, 14, 16])
@pytest.mark.parametrize("h", [256, 200, 33])
def test_split_encode_every_width_and_codebook_size(rq, oracle, sub, h):
    rng = np.random.default_rng(100 * sub + h)
    m, n = 4, 4_099                        # ragged last tile
    X = rng.standard_normal((n, m * sub)).astype(np.float32) * 3
    C = [rng.standard_normal((h, sub)).astype(np.float32) * 3 for _ in range(m)]
    got, old, ref = _enc_both(rq, oracle, X, C, m, h)
    assert np.array_equal(got, ref) and np.array_equal(old, ref)


@pytest.mark.parametrize("sub", [2, 4, 6, 8, 10, 12, 14, 16])
def test_split_encode_more_sub_quantizers_than_one_launch_holds(rq, oracle, sub):
    """every instantiated width with h < 256 and with MORE sub-quantizers than one launch group holds in LDS (gmax = 6 at
    sub > 8, 9 below: m = 12 runs two i0 groups of the filter and of the exact pass), on clustered small-integer data that sends
    a visible share of the pairs through the exact pass; ALL assignments against the oracle, both encode paths."""
    rng = np.random.default_rng(7000 + sub)
    m, h, n = 12, 200, 20_011
    cent = rng.integers(0, 24, (64, m * sub)).astype(np.float32)
    X = cent[rng.integers(0, 64, n)] + rng.integers(-2, 3, (n, m * sub)).astype(np.float32)
    C = [np.ascontiguousarray(X[rng.choice(n, h, replace=False)][:, i * sub:(i + 1) * sub]) for i in range(m)]
    rq.set_tuning("ENC_STATS", 1)
    try:
        got, old, ref = _enc_both(rq, oracle, X, C, m, h)
    finally:
        rq.set_tuning("ENC_STATS", 0)
    assert np.array_equal(got, ref) and np.array_equal(old, ref)


@pytest.mark.parametrize("waves", [8, 12, 16])
@pytest.mark.parametrize("case", ["ties", "dups", "exact_hits", "tiny", "huge", "mixed_scale", "negative_w"])
def test_split_encode_hostile_inputs(rq, oracle, case, waves):
    """What the filter's margin has to survive: massive exact ties (small-integer data: several centroids at the very
    same distance, in different 32-centroid tiles -> tiles re-run at the end, first index must win), duplicated
    centroids, vectors that ARE centroids (clamped zeros), magnitudes where the bound is unusable (every centroid is then
    evaluated exactly), wide dynamic range inside one sub-space.  (Distances that overflow to inf / NaN are outside every
    contract: Julia's `max(NaN, 0)` is NaN and `update_assignments!` then keeps centre 1 whenever `dmat[1, j]` is NaN, the C
    oracle's fmaxf returns 0 -- the kernels are only required to agree with the oracle on finite distances.)"""
    rng = np.random.default_rng({"ties": 1, "dups": 2, "exact_hits": 3, "tiny": 4, "huge": 5, "mixed_scale": 6, "negative_w": 7}[case])
    m, sub, h, n = 8, 16, 256, 6_000
    if case == "ties":
        X = rng.integers(0, 3, (n, m * sub)).astype(np.float32)
        C = [rng.integers(0, 3, (h, sub)).astype(np.float32) for _ in range(m)]
    elif case == "dups":
        X = rng.standard_normal((n, m * sub)).astype(np.float32)
        C = []
        for _ in range(m):
            c = rng.standard_normal((h, sub)).astype(np.float32)
            c[rng.permutation(h)[:h // 2]] = c[rng.integers(0, h, h // 2)]       # half of the rows are copies of other rows
            C.append(c)
    elif case == "exact_hits":
        C = [rng.standard_normal((h, sub)).astype(np.float32) * 50 for _ in range(m)]
        X = np.concatenate([C[i][rng.integers(0, h, n)] for i in range(m)], axis=1).astype(np.float32)
    elif case == "tiny":
        X = (rng.standard_normal((n, m * sub)) * 1e-14).astype(np.float32)
        C = [(rng.standard_normal((h, sub)) * 1e-14).astype(np.float32) for _ in range(m)]
    elif case == "huge":
        X = (rng.standard_normal((n, m * sub)) * 1e15).astype(np.float32)
        C = [(rng.standard_normal((h, sub)) * 1e15).astype(np.float32) for _ in range(m)]
    elif case == "mixed_scale":
        scale = np.exp(rng.uniform(-12, 12, (1, m * sub))).astype(np.float32)
        X = rng.standard_normal((n, m * sub)).astype(np.float32) * scale
        C = [rng.standard_normal((h, sub)).astype(np.float32) * scale[:, i * sub:(i + 1) * sub] for i in range(m)]
    else:       # far-away data: |c|^2 - 2<c, x> strongly negative, huge |x|^2 against small differences between centroids
        base = rng.standard_normal((1, m * sub)).astype(np.float32) * 1000
        X = base + rng.standard_normal((n, m * sub)).astype(np.float32)
        C = [base[:, i * sub:(i + 1) * sub] + rng.standard_normal((h, sub)).astype(np.float32) for i in range(m)]
    rq.set_tuning("ENC_SPLIT_WAVES", waves)
    try:
        got, old, ref = _enc_both(rq, oracle, np.ascontiguousarray(X), C, m, h)
    finally:
        rq.set_tuning("ENC_SPLIT_WAVES", 0)
    assert np.array_equal(old, ref), case
    assert np.array_equal(got, ref), (case, int((got != ref).sum()))


def test_split_encode_deep_shape_hostile(rq, oracle):
    """sub = 6 (two MFMAs per tile, hi and lo pieces packed into one K = 16 fragment), m = 16, ties and near-ties"""
    rng = np.random.default_rng(61)
    m, sub, h, n = 16, 6, 256, 5_000
    for scale, integer in ((1.0, False), (1.0, True), (1e-13, False)):
        if integer:
            X = rng.integers(0, 4, (n, m * sub)).astype(np.float32)
            C = [rng.integers(0, 4, (h, sub)).astype(np.float32) for _ in range(m)]
        else:
            X = (rng.standard_normal((n, m * sub)) * scale).astype(np.float32)
            C = [(rng.standard_normal((h, sub)) * scale).astype(np.float32) for _ in range(m)]
        got, old, ref = _enc_both(rq, oracle, X, C, m, h)
        assert np.array_equal(old, ref) and np.array_equal(got, ref), (scale, integer, int((got != ref).sum()))


def test_share_of_pairs_that_take_the_exact_pass(rq):
    """tuning ENC_STATS: how many (vector, sub-quantizer) pairs the bf16 filter leaves to the exact pass -- the figure the encode's
    speed rests on (2-3 % on SIFT-like, under 1 % on Deep-like data); everything on degenerate input."""
    import ctypes as C
    import torch
    import rayuela_jl_amd.synth as synth
    from rayuela_jl_amd import device as rqd, _lib

    def share(X, Cl, m):
        rq.set_tuning("ENC_STATS", 1)
        try:
            rqd.encode_pq(torch.from_numpy(X).cuda(), torch.from_numpy(synth.cat_codebooks(Cl)).cuda(), m, 256)
            torch.cuda.synchronize()
            out = (C.c_uint64 * 2)()
            _lib.check(_lib.lib().rq_last_encode_stats(C.cast(out, C.c_void_p)))
        finally:
            rq.set_tuning("ENC_STATS", 0)
        assert out[0] == X.shape[0] * m
        return out[1] / out[0]
    Xs = synth.sift_like(200_000, 128, seed=3)
    Cs = synth.codebooks(Xs[:20000], 8, 256, seed=4, iters=3, sample=20000)
    Xd = synth.deep_like(200_000, 96, seed=5)
    Cd = synth.codebooks(Xd[:20000], 16, 256, seed=6, iters=3, sample=20000)
    fs, fd = share(Xs, Cs, 8), share(Xd, Cd, 16)
    print("pairs left to the exact pass: sift-like %.3f %%, deep-like %.3f %%" % (100 * fs, 100 * fd))
    assert 0 < fs < 0.06 and 0 <= fd < 0.03
    Xz = np.zeros((5000, 128), dtype=np.float32)          # |x|^2 + max|c|^2 may be fine, but all-equal data ties everywhere
    Cz = [np.zeros((256, 16), dtype=np.float32) for _ in range(8)]
    assert share(Xz, Cz, 8) == 1.0


def test_encode_in_pieces_of_rows_equals_one_piece(rq, oracle):
    """tuning ENC_CHUNK_ROWS: the filter + exact-pass launches run per piece of rows (bounded scratch); codes must not depend on it"""
    import torch
    import rayuela_jl_amd.synth as synth
    from rayuela_jl_amd import device as rqd
    n, d, m = 50_011, 128, 8
    X = synth.sift_like(n, d, seed=21)
    C = synth.codebooks(X[:8000], m, 256, seed=22, iters=2, sample=8000)
    Ccat = synth.cat_codebooks(C)
    ref = oracle.encode_pq(X, Ccat, m, 256)
    Xd, Cd = torch.from_numpy(X).cuda(), torch.from_numpy(Ccat).cuda()
    rq.set_tuning("ENC_CHUNK_ROWS", 4096)
    try:
        got = rqd.encode_pq(Xd, Cd, m, 256).cpu().numpy()
    finally:
        rq.set_tuning("ENC_CHUNK_ROWS", 1 << 22)
    assert np.array_equal(got, ref)
