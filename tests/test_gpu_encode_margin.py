"""The bf16 matrix-core FILTER of encode_pq_split_kernel on evidence (VERDICT r3 Next #2).

The kernel's codes are exact because (csrc/rq_encode.hip, header of encode_pq_split_kernel)
    |(W_k + |x|^2) - v_k|  <=  e_k = 2^-14 (|c_k|^2 + |x|^2)        for every centroid k
where W_k is what the bf16 MFMAs produce and v_k the canonical distance (oracle/rq_oracle.c:264-328 <-> src/PQ.jl:40-41),
so the canonical argmin is always among {W_k <= min W + DELTA}, DELTA = 3 * 2^-14 (max|c|^2 + |x|^2).  Round 3 DERIVED e_k
from an assumed error model of the MFMA's internal accumulation.  Here it is MEASURED:
  * the kernel stores the very W values it filters on (rq_dev_encode_pq_filter_w) and the test takes the maximum of
    |(W_k + sb) - v_k| / (sa_k + sb) over hostile inputs and the 1e6 bench vectors (all 2e9 centroid distances);
  * the margin's numerator (tuning ENC_SPLIT_DELTA_MILLI, 3000 = shipped) is walked down until codes change: the ratio
    shipped / first-failing is the real safety factor;
  * all 8e6 / 1.6e7 assignments of the two bench shapes are compared with the oracle, not a sample.
The observed numbers are printed (pytest -s) and quoted in DESIGN.md section 4.2."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

E_K = 2.0 ** -14


def _ratio_max(rq, oracle, X, C, m, h, chunk=125_000):
    """max over (vector, sub-quantizer, centroid) of |(W + sb) - v| / (sa + sb), and the codes check"""
    import torch
    from rayuela_jl_amd import device as rqd
    n, d = X.shape
    sub = d // m
    Ccat = np.concatenate([np.ascontiguousarray(c, dtype=np.float32).reshape(-1) for c in C])
    Cd = torch.from_numpy(Ccat).cuda()
    C64 = torch.from_numpy(np.stack(C)).cuda().double()                       # [m][h][sub]
    sa = (C64 * C64).sum(2)                                                   # [m][h]
    worst, worst_unclamped, nbad = 0.0, 0.0, 0
    for a in range(0, n, chunk):
        Xc = np.ascontiguousarray(X[a:a + chunk])
        Xd = torch.from_numpy(Xc).cuda()
        codes, W = rqd.encode_pq_filter_w(Xd, Cd, m, h)
        assert not bool(torch.isnan(W).any()), "the split kernel did not run (W untouched)"
        U = torch.from_numpy(oracle.pq_distmat(Xc, Ccat, m, h)).cuda().double()     # [n][m][h] canonical, unclamped
        ref = oracle.encode_pq(Xc, Ccat, m, h)
        nbad += int((codes.cpu().numpy() != ref).sum())
        X64 = Xd.double().view(-1, m, sub)
        sb = (X64 * X64).sum(2)                                               # [n][m]
        den = sa[None, :, :] + sb[:, :, None]
        est = W.double() + sb[:, :, None]
        ok = den > 0
        r = ((est - U.clamp(min=0)).abs() / den)[ok]
        ru = ((est - U).abs() / den)[ok]
        worst = max(worst, float(r.max()))
        worst_unclamped = max(worst_unclamped, float(ru.max()))
        del U, W, est, den, r, ru
    return worst, worst_unclamped, nbad


def _hostile(case, rng, m, sub, h, n):
    if case == "ties":
        X = rng.integers(0, 3, (n, m * sub)).astype(np.float32)
        C = [rng.integers(0, 3, (h, sub)).astype(np.float32) for _ in range(m)]
    elif case == "dups":
        X = rng.standard_normal((n, m * sub)).astype(np.float32)
        C = []
        for _ in range(m):
            c = rng.standard_normal((h, sub)).astype(np.float32)
            c[rng.permutation(h)[:h // 2]] = c[rng.integers(0, h, h // 2)]
            C.append(c)
    elif case == "exact_hits":
        C = [rng.standard_normal((h, sub)).astype(np.float32) * 50 for _ in range(m)]
        X = np.concatenate([C[i][rng.integers(0, h, n)] for i in range(m)], axis=1).astype(np.float32)
    elif case == "mixed_scale":
        scale = np.exp(rng.uniform(-12, 12, (1, m * sub))).astype(np.float32)
        X = rng.standard_normal((n, m * sub)).astype(np.float32) * scale
        C = [rng.standard_normal((h, sub)).astype(np.float32) * scale[:, i * sub:(i + 1) * sub] for i in range(m)]
    elif case == "negative_w":     # far-away data: |c|^2 - 2<c, x> strongly negative, huge |x|^2 against small differences
        base = rng.standard_normal((1, m * sub)).astype(np.float32) * 1000
        X = base + rng.standard_normal((n, m * sub)).astype(np.float32)
        C = [base[:, i * sub:(i + 1) * sub] + rng.standard_normal((h, sub)).astype(np.float32) for i in range(m)]
    elif case == "cancel":         # x ~ c for many k at once with large norms: the worst cancellation in sa - 2g
        c0 = rng.standard_normal((1, sub)).astype(np.float32) * 300
        C = [(c0 + rng.standard_normal((h, sub)).astype(np.float32) * 0.05).astype(np.float32) for _ in range(m)]
        X = np.concatenate([c0 + rng.standard_normal((n, sub)).astype(np.float32) * 0.05 for _ in range(m)], axis=1).astype(np.float32)
    elif case == "bf16_edge":      # mantissas just below a bf16 rounding boundary: the largest split residuals
        def edge(shape):
            v = rng.standard_normal(shape).astype(np.float32)
            u = v.view(np.uint32)
            u = (u & np.uint32(0xFFFF0000)) | np.uint32(0x00007FFF)
            return u.view(np.float32)
        X = edge((n, m * sub))
        C = [edge((h, sub)) for _ in range(m)]
    else:
        raise ValueError(case)
    return np.ascontiguousarray(X), C


@pytest.mark.parametrize("sub", [16, 6])
@pytest.mark.parametrize("case", ["ties", "dups", "exact_hits", "mixed_scale", "negative_w", "cancel", "bf16_edge"])
def test_filter_error_within_bound_on_hostile_inputs(rq, oracle, case, sub):
    rng = np.random.default_rng(sum(map(ord, case)) + sub)
    m, h, n = (8, 256, 6_000) if sub == 16 else (16, 256, 5_000)
    X, C = _hostile(case, rng, m, sub, h, n)
    worst, worst_u, nbad = _ratio_max(rq, oracle, X, C, m, h)
    print("filter error %-12s sub=%2d: max |(W+sb)-v|/(sa+sb) = %.3e = %.3f * 2^-14 (unclamped %.3f * 2^-14)" % (
        case, sub, worst, worst / E_K, worst_u / E_K))
    assert nbad == 0, (case, sub, nbad)
    assert worst <= E_K, (case, sub, worst / E_K)


@pytest.mark.parametrize("shape", ["sift", "deep"])
def test_filter_error_and_margin_walk_at_full_size(rq, oracle, shape):
    """1e6 bench vectors: (a) the error of the filter values of a 250 000-row stratified sample against the bound, (b) ALL
    8e6 / 1.6e7 assignments against the oracle at the shipped margin, (c) the margin walked down until codes change."""
    import torch
    import rayuela_jl_amd.synth as synth
    import rayuela_jl_amd.synth_torch as st
    from rayuela_jl_amd import device as rqd
    n, h = 1_000_000, 256
    d, m = (128, 8) if shape == "sift" else (96, 16)
    dev = torch.device("cuda", 0)
    gen = (lambda rows, row0: st.sift_like(rows, d, seed=synth.SEED_BASE, ncentres=65536, row0=row0, device=dev)) if shape == "sift" \
        else (lambda rows, row0: st.deep_like(rows, d, seed=synth.SEED_BASE, row0=row0, device=dev))
    X = torch.cat([gen(250_000, o) for o in range(0, n, 250_000)], 0)
    S = gen(20_000, 3_100_000_000)
    if shape == "deep":        # BASELINE config 4 encodes the ROTATED base
        R = torch.from_numpy(synth.rotation(d)).to(dev)
        X, S = rqd.rotate_T(R, X), rqd.rotate_T(R, S)
    C = synth.codebooks(S.cpu().numpy(), m, h, seed=synth.SEED_CODEBOOK, iters=5, sample=20000)
    Xh = X.cpu().numpy()
    # (a) on every 4th 62500-row block (250 000 rows, 5e8 / 1e9 values): the host side of this sweep -- the oracle's h x n
    # distance matrix -- is what made the round-4 suite take ten minutes on the driver's box; round 4 ran all 1e6 rows and
    # found the maximum (0.053 / 0.363 * 2^-14) well inside the first block.  (b) below still covers EVERY assignment.
    na = 250_000
    Xa = np.ascontiguousarray(np.concatenate([Xh[o:o + na // 4] for o in range(0, n, n // 4)], 0))
    worst, worst_u, nbad = _ratio_max(rq, oracle, Xa, C, m, h)
    print("filter error %s shape, %d x %d x %d values: max = %.3e = %.3f * 2^-14 (unclamped %.3f * 2^-14); wrong codes %d of %d" % (
        shape, na, m, h, worst, worst / E_K, worst_u / E_K, nbad, na * m))
    assert nbad == 0
    assert worst <= E_K, worst / E_K       # (a)
    # (c) the margin walk: DELTA = numerator * 2^-14 * (max|c|^2 + |x|^2)
    Ccat = synth.cat_codebooks(C)
    ref = torch.from_numpy(oracle.encode_pq(Xh, Ccat, m, h)).to(dev)
    Cd = torch.from_numpy(Ccat).to(dev)
    first_bad, table = None, []
    try:
        for milli in (3000, 1500, 750, 375, 188, 94, 47, 23, 12, 6, 0):
            rq.set_tuning("ENC_SPLIT_DELTA_MILLI", milli)
            bad = int((rqd.encode_pq(X, Cd, m, h) != ref).sum())
            table.append((milli / 1000.0, bad))
            if bad and first_bad is None:
                first_bad = milli
    finally:
        rq.set_tuning("ENC_SPLIT_DELTA_MILLI", 3000)
    print("margin walk %s shape (numerator of DELTA_REL, codes that differ from the oracle of %d): %s" % (shape, n * m, table))
    print("   -> first failing numerator %s: safety factor of the shipped 3.0 >= %s" % (
        None if first_bad is None else first_bad / 1000.0, "inf" if not first_bad else "%.0fx" % (3000.0 / first_bad)))
    assert table[0][1] == 0              # (b) every assignment of the 1e6 vectors, not a sample
    # the shipped margin holds with a factor of at least 4 to spare on this data
    assert first_bad is None or first_bad < 750, table
