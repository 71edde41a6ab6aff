"""GPU parity of quantize_rvq (src/RVQ.jl:18-66) through the C ABI: codes, per-centre
counts and the final residual are bit-identical to the oracle (canonical fmaf-chain order)."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["rvq_sift_mini", "rvq_deep_mini"])
def test_quantize_rvq_matches_golden(rq, name):
    g = golden(name)
    C = [g["C"][i] for i in range(g["C"].shape[0])]
    codes, counts, Xr = rq.quantize_rvq_u8(g["X"], C, with_extras=True)
    assert np.array_equal(codes, g["codes"])
    assert np.array_equal(counts, g["counts"])
    assert np.array_equal(Xr.view(np.uint32), g["Xr"].view(np.uint32))
    B, singletons = rq.quantize_rvq(g["X"], C)
    assert B.dtype == np.int16 and np.array_equal(B, g["codes"].astype(np.int16) + 1)   # src/RVQ.jl:60-62
    for i in range(len(C)):
        n_unused = int((g["counts"][i] == 0).sum())
        assert (singletons[i] is None) == (n_unused == 0)
        if n_unused:
            assert singletons[i].shape == (n_unused, g["X"].shape[1])


@pytest.mark.parametrize("n,d,m,h,kind", [
    (20_000, 128, 8, 256, "sift"),
    (5_000, 96, 4, 256, "deep"),
    (3_001, 64, 5, 77, "deep"),      # h not a multiple of 32, ragged last tile
    (1_000, 30, 3, 64, "sift"),      # d % 4 != 0 -> scalar residual kernel, LDS-staged encode
    (33, 16, 2, 16, "deep"),
    (2_000, 256, 3, 256, "deep"),    # d > 128: chunked wide encode per stage
])
def test_quantize_rvq_vs_oracle_random(rq, oracle, n, d, m, h, kind):
    import rayuela_jl_amd.synth as synth
    X = synth.sift_like(n, d, seed=n) if kind == "sift" else synth.deep_like(n, d, seed=n)
    C = synth.rvq_codebooks(X, m, h, seed=n + 1, iters=1, sample=min(n, 2048))
    c0, cnt0, r0 = oracle.encode_rvq(X, C, with_extras=True)
    c1, cnt1, r1 = rq.quantize_rvq_u8(X, [C[i] for i in range(m)], with_extras=True)
    assert np.array_equal(c0, c1)
    assert np.array_equal(cnt0, cnt1)
    assert np.array_equal(r0.view(np.uint32), r1.view(np.uint32))


def test_device_resident_rvq_and_lsq_search(rq, oracle):
    """quantize_rvq on resident tensors, then the additive-quantizer scan on its codes (what experiment_rvq
    does, src/RVQ.jl:158-175): the scan is exact and ranks the nearest reconstruction first."""
    import torch
    import rayuela_jl_amd.synth as synth
    from rayuela_jl_amd import device as rqd
    n, d, m, h, nq, K = 30_000, 64, 4, 256, 20, 50
    X = synth.deep_like(n + nq, d, seed=77)
    Xb, Xq = X[:n], X[n:]
    C = synth.rvq_codebooks(Xb, m, h, seed=78, iters=3)
    Xr = torch.from_numpy(Xb.copy()).cuda()
    codes, counts = rqd.encode_rvq(Xr, torch.from_numpy(C).cuda(), want_counts=True)
    c0, cnt0, r0 = oracle.encode_rvq(Xb, C, with_extras=True)
    assert np.array_equal(codes.cpu().numpy(), c0)
    assert np.array_equal(counts.cpu().numpy().astype(np.uint32), cnt0)
    assert np.array_equal(Xr.cpu().numpy().view(np.uint32), r0.view(np.uint32))
    # database norms of the reconstructions, then linscan_lsq (src/Linscan.jl:118-157)
    recon = np.zeros((n, d), dtype=np.float32)
    for i in range(m):
        recon += C[i][c0[:, i]]
    dbnorms = (recon.astype(np.float64) ** 2).sum(1).astype(np.float32)
    dists, idx = rq.linscan_lsq(c0, Xq, [C[i] for i in range(m)], dbnorms, np.eye(d, dtype=np.float32), K)
    d_or, i_or = oracle.linscan_lsq(c0, C.reshape(m * h, d), Xq, dbnorms, K)
    assert np.array_equal(idx, i_or)
    assert np.array_equal(dists.view(np.uint32), d_or.view(np.uint32))
    # the f64 nearest RECONSTRUCTION must lead the f32 ranking (up to rounding ties among the first few)
    score = (recon.astype(np.float64) ** 2).sum(1)[None, :] - 2.0 * Xq.astype(np.float64) @ recon.astype(np.float64).T
    best = score.argmin(1) + 1
    assert (idx[:, :5] == best[:, None].astype(idx.dtype)).any(axis=1).all()
    assert (idx[:, 0] == best.astype(idx.dtype)).mean() >= 0.9


def test_train_rvq_contract(rq, oracle):
    """train_rvq (src/RVQ.jl:86-127): every stage lowers the error, more Lloyd iterations lower it further,
    the returned codes are quantize_rvq of the returned codebooks and the reported error is qerror(X, B, C)."""
    import rayuela_jl_amd.synth as synth
    X = synth.sift_like(20_000, 64, seed=21)
    errs = []
    for m in (1, 2, 4):
        C, B, e = rq.train_rvq(X, m, 64, niter=8, seed=3)
        errs.append(e)
        assert len(C) == m and C[0].shape == (64, 64) and B.shape == (20_000, m) and B.dtype == np.int16
    assert errs[0] > errs[1] > errs[2] > 0
    C1, B1, e1 = rq.train_rvq(X, 4, 64, niter=1, seed=3)
    assert errs[2] < e1
    C, B, e = rq.train_rvq(X, 4, 64, niter=8, seed=3)
    # same seed, same bits: the training loop is deterministic since round 2 (fixed-order segment sums)
    assert e == errs[2]
    Bq, singles = rq.quantize_rvq(X, C)
    assert np.array_equal(Bq, B)
    recon = np.zeros(X.shape, dtype=np.float64)
    for i in range(4):
        recon += C[i][B[:, i].astype(np.int64) - 1]
    e64 = ((X.astype(np.float64) - recon) ** 2).sum() / X.shape[0]
    assert abs(e - e64) <= 1e-5 * e64
    # the oracle encode agrees on the trained codebooks too
    assert np.array_equal(oracle.encode_rvq(X, np.stack(C)).astype(np.int16) + 1, B)


def test_experiment_rvq_end_to_end(rq):
    """experiment_rvq (src/RVQ.jl:130-175) on synthetic data: train -> norms codebook -> encode the base ->
    linscan_lsq -> recall; 32 bits of RVQ must find most true neighbours of clustered data in the top 50."""
    import rayuela_jl_amd.synth as synth
    from rayuela_jl_amd.experiments import experiment_rvq
    d, m, h, knn = 32, 4, 256, 50
    Xb = synth.sift_like(20_000, d, seed=5)
    Xq = synth.sift_like(64, d, seed=6)
    dd = ((Xq.astype(np.float64)[:, None, :] - Xb.astype(np.float64)[None, :, :]) ** 2).sum(-1)
    gt = (dd.argmin(1) + 1).astype(np.uint32)
    C, B, train_error, B_base, recall = experiment_rvq(Xb[:8000], Xb, Xq, gt, m, h, niter=6, knn=knn)
    assert recall.shape == (knn,) and (np.diff(recall) >= 0).all() and recall[-1] > 0.5
    assert train_error > 0 and B_base.shape == (20_000, m)
