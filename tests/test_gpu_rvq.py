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
