"""GPU tests of the training reductions and loops.  Floating-point reductions in a
different order than the CPU loops: tolerance 1e-5 relative for the reductions, 1e-4 on the objective
curve (stated here, per the north star's float bar); reconstruction is exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(n=20000, d=64, m=8, h=64, seed=0):
    import torch
    import rayuela_jl_amd.synth as synth
    X = synth.sift_like(n, d, seed=seed + 1)
    C = synth.codebooks(X, m, h, seed=seed + 2, iters=1, sample=2000)
    return X, C


def test_reductions_match_float64(rq, oracle):
    import torch
    import rayuela_jl_amd.synth as synth
    from rayuela_jl_amd import device as rqd
    from oracle import train_oracle as to
    X, C = _setup()
    n, d = X.shape
    m, h = 8, 64
    off = to.offsets(d, m)
    codes = oracle.encode_pq(X, synth.cat_codebooks(C), m, h)
    Xd, cd = torch.from_numpy(X).cuda(), torch.from_numpy(codes).cuda()
    Ccat = torch.from_numpy(synth.cat_codebooks(C)).cuda()
    # reconstruct: exact
    CB = rqd.reconstruct(cd, Ccat, d, h)
    CB0 = to.reconstruct(C, codes, off, d)
    assert np.array_equal(CB.cpu().numpy(), CB0)
    # qerror
    e0 = ((X.astype(np.float64) - CB0) ** 2).sum() / n
    assert abs(rqd.qerror(Xd, CB) - e0) <= 1e-9 * e0
    # gram
    G0 = X.astype(np.float64).T @ CB0.astype(np.float64)
    G = rqd.gram(Xd, CB).cpu().numpy()
    assert np.allclose(G, G0, rtol=1e-5, atol=1e-5 * np.abs(G0).max())
    # update_centers (+ counts, + an artificially empty cluster keeps its centre)
    codes2 = codes.copy()
    codes2[codes2[:, 0] == 5, 0] = 6
    C2 = torch.from_numpy(synth.cat_codebooks(C)).cuda()
    counts = rqd.update_centers(C2, Xd, torch.from_numpy(codes2).cuda(), m, h).cpu().numpy()
    Cn = to.update_centers(C, X, codes2, off, h)
    got = C2.cpu().numpy()
    assert np.allclose(got, synth.cat_codebooks(Cn), rtol=1e-5, atol=1e-4)
    assert counts[0, 5] == 0 and counts.sum() == n * m
    assert np.array_equal(counts, np.stack([np.bincount(codes2[:, i], minlength=h) for i in range(m)]))


def test_train_opq_follows_the_oracle_loop(rq, oracle):
    import rayuela_jl_amd.synth as synth
    from oracle import train_oracle as to
    X, C0 = _setup(n=12000, d=32, m=4, h=32, seed=3)
    R0 = synth.rotation(32, seed=5)
    C0r = synth.codebooks(oracle.rotate_T(R0, X), 4, 32, seed=9, iters=0, sample=2000)
    niter = 4
    C, B, R, obj = rq.train_opq(X, 4, 32, niter, "natural", R0=R0, C0=C0r)          # C ABI (rq_train_opq)
    Co, codes_o, Ro, obj_o = to.train_opq(X, 4, 32, niter, R0, C0r)
    # the device-resident python loop over the rq_dev_* pieces takes the same steps
    from rayuela_jl_amd import train as tr
    C2, B2, R2, obj2 = tr.train_opq(X, 4, 32, niter, "natural", R0=R0, C0=C0r)
    # the two front ends take their polar factor from different SVDs (host Jacobi in the library, LAPACK in numpy),
    # so R -- and with it a handful of near-tie assignments -- agree to float tolerance; each front end by
    # itself is bit-reproducible (test_training_is_bit_reproducible)
    assert np.allclose(obj2, obj, rtol=1e-4) and np.abs(R2 - R).max() < 2e-2
    assert obj.shape == (niter + 1,)
    assert np.allclose(obj, obj_o, rtol=3e-4)
    assert (np.diff(obj) <= 1e-5 * obj[:-1]).all()          # the alternating minimisation never goes up
    assert np.abs(R @ R.T - np.eye(32)).max() < 1e-5         # R stays orthonormal
    assert (B - 1 != codes_o).mean() < 2e-2                  # same assignments up to float near-ties


def test_train_pq_reduces_the_error(rq, oracle):
    import rayuela_jl_amd.synth as synth
    X, _ = _setup(n=30000, d=64, m=8, h=64, seed=7)
    C1, B1, e1 = rq.train_pq(X, 8, 64, niter=1, seed=1)
    C, B, e = rq.train_pq(X, 8, 64, niter=12, seed=1)
    assert e < e1
    assert B.dtype == np.int16 and B.min() >= 1 and B.max() <= 64
    # the returned codes are the quantisation of X by the returned codebooks (quantize_pq contract)
    assert np.array_equal(B, rq.quantize_pq(X, C))
    # and the reported error is qerror_pq of exactly that pair
    from oracle import train_oracle as to
    CB = to.reconstruct(C, (B - 1).astype(np.uint8), to.offsets(64, 8), 64)
    e0 = ((X.astype(np.float64) - CB) ** 2).sum() / X.shape[0]
    assert abs(e - e0) <= 1e-6 * e0


@pytest.mark.parametrize("kind", ["sift", "clustered"])
def test_train_pq_lands_where_clustering_kmeans_rules_land(rq, oracle, kind):
    """rq_train_pq against oracle/train_oracle.py::train_pq_clustering -- Clustering.kmeans' own rules (kmeans++ seeding,
    repick of emptied centres proportional to cost, stop on |objv - prev| < 1e-6) -- on 3 seeds each: the draws differ
    (different random streams on the two sides), so the comparison is the final quantisation error, within 0.5 % on the
    seed average and 1.5 % seed by seed.  `clustered`: fewer natural clusters than centres -- the shape that empties
    clusters -- so that the repick rule is exercised on both sides."""
    import rayuela_jl_amd.synth as synth
    from oracle import train_oracle as to
    if kind == "sift":
        X = synth.sift_like(20000, 32, seed=77)
        m, h, niter = 4, 64, 25
    else:
        rng = np.random.default_rng(9)
        cen = rng.uniform(-200, 200, (24, 16)).astype(np.float32)
        X = (cen[rng.integers(0, 24, 12000)] + rng.standard_normal((12000, 16)) * 2.0).astype(np.float32)
        m, h, niter = 2, 64, 25
    eg, eo = [], []
    for seed in (1, 2, 3):
        C, B, e = rq.train_pq(X, m, h, niter=niter, seed=seed)
        assert np.array_equal(B, rq.quantize_pq(X, C))
        _, _, e0 = to.train_pq_clustering(X, m, h, niter, seed)
        eg.append(e); eo.append(e0)
    print("train_pq %s: library %s  oracle (Clustering rules) %s" % (kind, np.round(eg, 3), np.round(eo, 3)))
    assert abs(np.mean(eg) - np.mean(eo)) <= 0.005 * np.mean(eo), (eg, eo)
    assert all(abs(a - b) <= 0.015 * b for a, b in zip(eg, eo)), (eg, eo)


def test_experiment_opq_end_to_end(rq, oracle):
    """experiment_opq (src/OPQ.jl:142-171) on synthetic data: train -> quantize_opq -> linscan_opq ->
    eval_recall, all through the C ABI; the search leg must equal the oracle given the trained model."""
    import rayuela_jl_amd.synth as synth
    from rayuela_jl_amd.experiments import experiment_opq
    d, m, h, knn = 32, 4, 256, 50
    Xb = synth.deep_like(20000, d, seed=1)
    Xt = Xb[:8000]
    Xq = synth.deep_like(64, d, seed=2)
    dd = ((Xq.astype(np.float64)[:, None, :] - Xb.astype(np.float64)[None, :, :]) ** 2).sum(-1)
    gt = (dd.argmin(1) + 1).astype(np.uint32)                       # one-based like the ivecs ground truth
    C, B, R, obj, B_base, recall = experiment_opq(Xt, Xb, Xq, gt, m, h, "natural", niter=5, knn=knn)
    assert recall.shape == (knn,) and (np.diff(recall) >= 0).all() and recall[-1] > 0.5
    assert (np.diff(obj) <= 1e-5 * obj[:-1]).all()
    # search parity for the trained model: oracle encode + oracle scan on the rotated data
    codes0 = oracle.encode_opq(Xb, R, synth.cat_codebooks(C), m, h)
    assert np.array_equal(B_base, codes0.astype(np.int16) + 1)
    d0, i0 = oracle.linscan_aqd_query(codes0, np.stack(C), oracle.rotate_T(R, Xq), knn)
    rec0 = oracle.eval_recall(gt, i0 + 1, knn)
    assert np.allclose(recall, rec0)


def test_train_opq_random_init_and_polar_factor(rq):
    """init = "random": R must come back orthonormal, the objective must not increase, and the host
    Jacobi polar factor must agree with LAPACK's SVD on the learned rotation."""
    import rayuela_jl_amd.synth as synth
    X = synth.deep_like(6000, 48, seed=11)
    C, B, R, obj = rq.train_opq(X, 6, 16, 6, "random", seed=5)
    assert np.abs(R @ R.T - np.eye(48)).max() < 1e-5
    assert (np.diff(obj) <= 1e-5 * obj[:-1]).all()
    # R' X re-encoded with the returned codebooks reproduces B (quantize_opq contract)
    assert np.array_equal(rq.quantize_opq(X, R, C), B)
    with pytest.raises(ValueError):
        rq.train_opq(X, 6, 16, 1, "bogus")


def test_training_is_bit_reproducible(rq):
    """Same inputs + same seed => identical bits, run after run: update_centers sums every (code, dimension) in
    ascending row order by ONE owner thread and combines the workgroup slices in fixed order, qerror uses a fixed
    reduction tree, the assignment kernels and the host Jacobi polar factor are deterministic."""
    import rayuela_jl_amd.synth as synth
    X = synth.sift_like(40000, 64, seed=21)
    a = rq.train_pq(X, 8, 256, niter=6, seed=3)
    b = rq.train_pq(X, 8, 256, niter=6, seed=3)
    assert all(np.array_equal(x, y) for x, y in zip(a[0], b[0])) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    Xd = synth.deep_like(30000, 48, seed=22)
    a = rq.train_opq(Xd, 6, 64, 5, "random", seed=9)
    b = rq.train_opq(Xd, 6, 64, 5, "random", seed=9)
    assert all(np.array_equal(x, y) for x, y in zip(a[0], b[0])) and np.array_equal(a[1], b[1])
    assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32)) and np.array_equal(a[3].view(np.uint32), b[3].view(np.uint32))
    a = rq.train_rvq(X, 3, 64, niter=4, seed=5)
    b = rq.train_rvq(X, 3, 64, niter=4, seed=5)
    assert all(np.array_equal(x, y) for x, y in zip(a[0], b[0])) and np.array_equal(a[1], b[1]) and a[2] == b[2]


def test_update_centers_any_width(rq):
    """Dimension chunks of 128: d = 200 (one full-dimensional codebook, RVQ-style) and d = 960, m = 8 (GIST-shape PQ)."""
    import torch
    from rayuela_jl_amd import device as rqd
    from oracle import train_oracle as to
    rng = np.random.default_rng(5)
    for n, d, m, h in [(5000, 200, 1, 256), (3000, 960, 8, 128)]:
        X = rng.standard_normal((n, d)).astype(np.float32)
        codes = rng.integers(0, h, (n, m)).astype(np.uint8)
        off = to.offsets(d, m)
        C = [rng.standard_normal((h, int(off[i + 1] - off[i]))).astype(np.float32) for i in range(m)]
        Ccat = torch.from_numpy(np.concatenate([c.reshape(-1) for c in C])).cuda()
        counts = rqd.update_centers(Ccat, torch.from_numpy(X).cuda(), torch.from_numpy(codes).cuda(), m, h).cpu().numpy()
        Cn = to.update_centers(C, X, codes, off, h)
        assert np.allclose(Ccat.cpu().numpy(), np.concatenate([c.reshape(-1) for c in Cn]), rtol=1e-5, atol=1e-5)
        assert np.array_equal(counts, np.stack([np.bincount(codes[:, i], minlength=h) for i in range(m)]))


def test_kmeanspp_seeding(rq):
    """init=:kmpp (src/PQ.jl:86): seeds are distinct rows of X, the returned centres are exactly their sub-vectors,
    D^2 sampling covers well-separated clusters (uniform row sampling does not), and the draw is reproducible."""
    rng = np.random.default_rng(3)
    ncl, per, d, m = 48, 400, 16, 2
    centres = rng.uniform(-1000, 1000, (ncl, d)).astype(np.float32)
    X = (centres[np.repeat(np.arange(ncl), per)] + rng.standard_normal((ncl * per, d)).astype(np.float32)).astype(np.float32)
    X = X[rng.permutation(X.shape[0])]
    seeds, C = rq.kmpp_seeds(X, m, ncl, seed=11)
    assert seeds.shape == (m, ncl) and seeds.min() >= 0 and seeds.max() < X.shape[0]
    sub = d // m
    for i in range(m):
        assert len(set(seeds[i].tolist())) == ncl                                   # distinct points
        assert np.array_equal(C[i], X[seeds[i], i * sub:(i + 1) * sub])             # centres ARE the chosen rows
        # which cluster does every seed come from?  D^2 sampling: (almost) one seed per cluster
        lab = ((X[seeds[i], None, i * sub:(i + 1) * sub] - centres[None, :, i * sub:(i + 1) * sub]) ** 2).sum(-1).argmin(1)
        assert len(set(lab.tolist())) >= ncl - 2
    s2, C2 = rq.kmpp_seeds(X, m, ncl, seed=11)
    assert np.array_equal(seeds, s2)
    s3, _ = rq.kmpp_seeds(X, m, ncl, seed=12)
    assert not np.array_equal(seeds, s3)
    # uniform row sampling of the same size leaves ~ncl/e clusters without a seed: the contrast the test relies on
    uni = rng.integers(0, X.shape[0], ncl)
    lab_u = ((X[uni, None, :sub] - centres[None, :, :sub]) ** 2).sum(-1).argmin(1)
    assert len(set(lab_u.tolist())) < ncl - 5
    # identical points: every cost is 0 after the first seed; the call must still return h valid rows
    Z = np.ones((500, 8), dtype=np.float32)
    sz, _ = rq.kmpp_seeds(Z, 1, 16, seed=1)
    assert sz.min() >= 0 and sz.max() < 500


def test_train_pq_with_kmeanspp_beats_uniform_seeding_on_clustered_data(rq):
    rng = np.random.default_rng(4)
    ncl, per, d = 64, 300, 32
    centres = rng.uniform(-500, 500, (ncl, d)).astype(np.float32)
    X = (centres[np.repeat(np.arange(ncl), per)] + rng.standard_normal((ncl * per, d))).astype(np.float32)
    X = X[rng.permutation(X.shape[0])]
    _, _, e_pp = rq.train_pq(X, 4, 64, niter=2, seed=5)
    rq.set_tuning("TRAIN_KMPP", 0)
    try:
        _, _, e_uni = rq.train_pq(X, 4, 64, niter=2, seed=5)
    finally:
        rq.set_tuning("TRAIN_KMPP", 1)
    assert e_pp < e_uni


def test_results_survive_an_hdf5_round_trip(rq, tmp_path):
    """train_opq -> save_results_opq (demos/experiment_utils.jl:28-38 layout) -> load -> the reloaded model encodes and
    searches exactly like the one in memory."""
    import rayuela_jl_amd.synth as synth
    h5 = rq.h5results
    if not h5.available():
        pytest.skip("libhdf5 not found")
    d, m, h, knn = 32, 4, 256, 20          # the scan is for h = 256 (uint8 codes)
    Xb = synth.deep_like(6000, d, seed=31)
    Xq = synth.deep_like(16, d, seed=32)
    C, B, R, obj = rq.train_opq(Xb[:3000], m, h, 3, "natural", seed=2)
    B_base = rq.quantize_opq(Xb, R, C)
    dists, idx = rq.linscan_opq(B_base, Xq, C, 8 * m, R, knn)
    path = str(tmp_path / "opq.h5")
    h5.save_results_opq(path, 1, C, B, R, obj[-1], B_base, np.linspace(0, 1, knn))
    C2, B2, R2, err = h5.load_opq(path, m, 1)
    assert np.array_equal(B2, B) and float(np.asarray(err)) == float(obj[-1])
    Bb2 = h5.h5read(path, "1/B_base")                       # zero-based UInt8: the scan's wire format, used as it is
    assert np.array_equal(rq.quantize_opq(Xb, R2, C2), B_base)
    d2, i2 = rq.linscan_opq(Bb2, Xq, C2, 8 * m, R2, knn)
    assert np.array_equal(i2, idx) and np.array_equal(d2.view(np.uint32), dists.view(np.uint32))


def _polar_ref(G):
    U, s, Vt = np.linalg.svd(G.astype(np.float64))
    return U @ Vt, s


def _polar_cases(d, rng):
    Q1, _ = np.linalg.qr(rng.standard_normal((d, d)))
    Q2, _ = np.linalg.qr(rng.standard_normal((d, d)))
    yield "gaussian", rng.standard_normal((d, d))
    yield "kappa1e6", Q1 @ np.diag(np.logspace(0, -6, d)) @ Q2
    yield "pca_like", (Q1 * (1.0 / (1.0 + np.arange(d)) ** 1.5)) @ Q1.T * 3.0e7      # a covariance: symmetric, fast decay, big norm
    yield "flat", Q1 * 1e-5                                                          # all singular values equal, tiny norm
    yield "orthogonal_already", Q2


@pytest.mark.parametrize("d", [2, 6, 30, 96, 128, 130, 200, 384, 500, 960])
def test_device_polar_factor_matches_the_svd(rq, d):
    """src/OPQ.jl:112-113: U, S, VV = svd(X * CB'); R = U * VV'.  The Newton-Schulz kernel (what rq_train_opq runs) and the
    Jacobi kernel against LAPACK's SVD in float64: |R - U V'| <= 1e-6 (f32 storage of R and of G; the factor's condition
    number is 1 / sigma_min, so the ill-conditioned case gets a tolerance scaled by it), R orthogonal to f32 rounding."""
    import torch
    from rayuela_jl_amd import device as rqd
    rng = np.random.default_rng(d)
    for name, G in _polar_cases(d, rng):
        if d > 256 and name not in ("gaussian", "pca_like"):      # (the 64 x 64-tile kernel; float64 SVDs of this size take a second each)
            continue
        G32 = np.ascontiguousarray(G, dtype=np.float32)
        P, s = _polar_ref(G32)
        tol = max(2e-6, 4e-7 * s[0] / s[-1] * 0.01) if name != "kappa1e6" else 5e-2
        for method in ([0, 1] if (d % 2 == 0 and d <= 128) else [0]):
            R, ok, steps = rqd.polar_factor(torch.from_numpy(G32).cuda(), method)
            assert ok, (name, d, method, steps)
            R = R.cpu().numpy().astype(np.float64)
            assert np.abs(R @ R.T - np.eye(d)).max() < 5e-7 * max(1, d ** 0.5), (name, d, method)
            assert np.abs(R - P).max() < tol, (name, d, method, steps, np.abs(R - P).max())
            if method == 0:
                assert steps <= (60 if name == "kappa1e6" else 30), (name, d, steps)


def test_device_polar_factor_gives_up_on_a_singular_matrix_and_training_falls_back(rq):
    """A rank-deficient G has no unique polar factor: the Newton-Schulz iteration cannot converge and says so (status 1);
    rq_train_opq then takes the Jacobi / host path (which completes the basis) -- a base with a constant dimension does that
    from the first iteration on."""
    import torch
    from rayuela_jl_amd import device as rqd
    from rayuela_jl_amd import _lib
    rng = np.random.default_rng(3)
    G = rng.standard_normal((64, 64)).astype(np.float32)
    G[:, 5] = 0.0
    R, ok, steps = rqd.polar_factor(torch.from_numpy(G).cuda(), 0)
    assert not ok
    R, ok, steps = rqd.polar_factor(torch.zeros((64, 64), device="cuda"), 0)
    assert not ok and steps == 0
    import rayuela_jl_amd.synth as synth
    X = synth.deep_like(5000, 32, seed=4)
    X[:, 7] = 0.0
    C, B, Rm, obj = rq.train_opq(X, 4, 16, 3, "natural", seed=1)
    prof = _lib.train_profile()
    assert prof["host_polar"] + prof["jacobi_sweeps"] > 0
    assert np.abs(Rm @ Rm.T - np.eye(32)).max() < 1e-5
    assert (np.diff(obj) <= 1e-5 * obj[:-1]).all()


def test_train_opq_newton_schulz_and_jacobi_agree(rq):
    import rayuela_jl_amd.synth as synth
    from rayuela_jl_amd import _lib
    X = synth.sift_like(30000, 64, seed=9)
    a = rq.train_opq(X, 8, 64, 5, "natural", seed=2)
    pa = _lib.train_profile()
    assert pa["ns_steps"] > 0 and pa["jacobi_sweeps"] == 0 and pa["host_polar"] == 0
    rq.set_tuning("TRAIN_GPU_POLAR", 2)
    try:
        b = rq.train_opq(X, 8, 64, 5, "natural", seed=2)
        pb = _lib.train_profile()
    finally:
        rq.set_tuning("TRAIN_GPU_POLAR", 1)
    assert pb["ns_steps"] == 0 and pb["jacobi_sweeps"] > 0
    assert np.abs(a[2] - b[2]).max() < 2e-3               # the rotations (chaotic over iterations: codes flip on 1e-7 differences)
    assert np.allclose(a[3], b[3], rtol=2e-4)             # the objective curves


@pytest.mark.parametrize("n,d,m,h", [(50_000, 128, 8, 256), (20_001, 96, 16, 256), (7_000, 64, 8, 64), (3_003, 30, 5, 17),
                                     (70_000, 200, 4, 256), (999, 8, 8, 256), (40_000, 130, 2, 100), (5_000, 40, 8, 64),
                                     (6_000, 64, 8, 200), (9_001, 50, 8, 256)])
def test_update_centers_kernels_agree(rq, n, d, m, h):
    """Two kernels for Clustering.update_centers! (call site src/OPQ.jl:121): the round-3 owner-thread scatter (sequential f32
    sums in row order) and the matrix-core kernel (one-hot x three exact bf16 pieces of x; the default), which adds the same
    f32 values in the MFMA's order: identical counts, centres within 1e-5 of float64 for both.  Narrow and mixed sub-space
    widths included (d / m = 96 / 16, 30 / 5, 8 / 8, 40 / 8, 64 / 8, 50 / 8)."""
    import torch
    from rayuela_jl_amd import device as rqd
    from oracle import train_oracle as to
    rng = np.random.default_rng(n + d)
    X = (rng.standard_normal((n, d)) * 30).astype(np.float32)
    X[::7] *= 1e-3                                           # mixed magnitudes: all three bf16 pieces matter
    codes = rng.integers(0, h, (n, m), dtype=np.uint8)
    codes[:, 0] = np.minimum(codes[:, 0], h // 2)            # leaves empty clusters: they keep their value
    off = to.offsets(d, m)
    C0 = rng.standard_normal(h * d).astype(np.float32)
    Xd, cd = torch.from_numpy(X).cuda(), torch.from_numpy(codes).cuda()
    outs = []
    for mfma in (0, 1):
        rq.set_tuning("TRAIN_CENTERS_MFMA", mfma)
        try:
            Cd = torch.from_numpy(C0.copy()).cuda()
            cnt = rqd.update_centers(Cd, Xd, cd, m, h)
            outs.append((Cd.cpu().numpy(), cnt.cpu().numpy()))
        finally:
            rq.set_tuning("TRAIN_CENTERS_MFMA", 1)
    assert np.array_equal(outs[0][1], outs[1][1])
    assert np.allclose(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-4)
    # against float64
    pos = 0
    for q in range(m):
        sub = off[q + 1] - off[q]
        Cq = C0[pos:pos + h * sub].reshape(h, sub).astype(np.float64).copy()
        cnt = np.bincount(codes[:, q], minlength=h)
        sums = np.zeros((h, sub))
        np.add.at(sums, codes[:, q], X[:, off[q]:off[q + 1]].astype(np.float64))
        Cq[cnt > 0] = sums[cnt > 0] / cnt[cnt > 0, None]
        for o in outs:
            got = o[0][pos:pos + h * sub].reshape(h, sub)
            assert np.allclose(got, Cq, rtol=1e-5, atol=1e-4), q
            assert np.array_equal(o[1][q], cnt)
        pos += h * sub


@pytest.mark.parametrize("n,d,m,h", [(30_000, 128, 8, 256), (20_001, 96, 16, 256), (20_001, 96, 8, 256), (5_000, 64, 8, 64), (12_345, 160, 4, 100),
                                     (8_000, 128, 32, 256), (6_000, 256, 32, 64), (4_000, 30, 5, 17), (3_000, 320, 8, 64),
                                     (2_500, 960, 8, 256), (2_000, 520, 4, 16),
                                     (9_000, 256, 8, 256), (7_777, 32, 2, 16), (100, 8, 2, 4), (31, 128, 8, 256)])
def test_gram_and_qerror_from_codes_match_the_reconstructed_forms(rq, n, d, m, h):
    """gram_codes / qerror_codes gather CB[j] = C[codes[j]] inside the kernel (src/OPQ.jl:101,108,112 without the n x d
    temporary): against float64 on the explicitly reconstructed CB, 1e-5 relative (f32 accumulation in another order)."""
    import torch
    from rayuela_jl_amd import device as rqd
    from oracle import train_oracle as to
    rng = np.random.default_rng(n + d + m)
    X = (rng.standard_normal((n, d)) * 20 + 3).astype(np.float32)
    codes = rng.integers(0, h, (n, m), dtype=np.uint8)
    off = to.offsets(d, m)
    C = [rng.standard_normal((h, off[q + 1] - off[q])).astype(np.float32) * 20 for q in range(m)]
    Ccat = np.concatenate([c.reshape(-1) for c in C])
    CB = to.reconstruct(C, codes, off, d)
    Xd, cd, Cd = torch.from_numpy(X).cuda(), torch.from_numpy(codes).cuda(), torch.from_numpy(Ccat).cuda()
    G0 = X.astype(np.float64).T @ CB.astype(np.float64)
    G = rqd.gram_codes(Xd, cd, Cd, h).cpu().numpy()
    assert np.allclose(G, G0, rtol=1e-5, atol=1e-5 * np.abs(G0).max())
    e0 = ((X.astype(np.float64) - CB) ** 2).sum() / n
    assert abs(rqd.qerror_codes(Xd, cd, Cd, h) - e0) <= 1e-9 * e0
    # and the reconstructed forms agree with them
    CBd = rqd.reconstruct(cd, Cd, d, h)
    assert np.allclose(rqd.gram(Xd, CBd).cpu().numpy(), G, rtol=1e-5, atol=1e-5 * np.abs(G0).max())
    assert abs(rqd.qerror(Xd, CBd) - e0) <= 1e-9 * e0


def test_codes_forms_refuse_shapes_without_aligned_subspaces(rq):
    import torch
    from rayuela_jl_amd import device as rqd
    X = torch.zeros((100, 30), device="cuda")                    # d = 30, m = 4: sub-spaces 8, 8, 7, 7 -- the last starts at 23
    codes = torch.zeros((100, 4), dtype=torch.uint8, device="cuda")
    C = torch.zeros((16 * 30,), device="cuda")
    with pytest.raises(Exception):
        rqd.gram_codes(X, codes, C, 16)
    with pytest.raises(Exception):
        rqd.qerror_codes(X, codes, C, 16)


def test_train_opq_on_a_shape_without_aligned_subspaces(rq):
    """d = 30, m = 4: sub-spaces of 8, 8, 7, 7 dimensions (src/utils.jl:179-203) -- the (codes, C) forms of gram / qerror do
    not apply, the loop materialises CB as in round 3; same objective curve as the torch front end on the same start."""
    import rayuela_jl_amd.synth as synth
    from rayuela_jl_amd import train as tr
    from oracle import train_oracle as to
    X = synth.deep_like(8000, 30, seed=31)
    rng = np.random.default_rng(0)
    off = to.offsets(30, 4)
    C0 = [X[rng.choice(8000, 16, replace=False)][:, off[i]:off[i + 1]].copy() for i in range(4)]
    R0 = np.eye(30, dtype=np.float32)
    C, B, R, obj = rq.train_opq(X, 4, 16, 4, "natural", R0=R0, C0=C0)
    C2, B2, R2, obj2 = tr.train_opq(X, 4, 16, 4, "natural", R0=R0, C0=C0)
    assert np.allclose(obj, obj2, rtol=3e-4)
    assert (np.diff(obj) <= 1e-5 * obj[:-1]).all()
    assert np.abs(R @ R.T - np.eye(30)).max() < 1e-5
    assert np.array_equal(rq.quantize_opq(X, R, C), B)
