"""Device-resident entry points (rq_dev_* of include/rayuela_hip.h) on torch CUDA tensors.

torch is plumbing here: it owns the device memory and the stream; every launch goes to the current
torch stream, so torch.cuda.Event timing brackets exactly the kernels of this library."""
import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, dtype, name):
    if not (t.is_cuda and t.is_contiguous() and t.dtype == dtype):
        raise TypeError("%s must be a contiguous CUDA tensor of dtype %s" % (name, dtype))
    return t.data_ptr()


def encode_pq(X, Ccat, m, h, out=None):
    n, d = X.shape
    out = torch.empty((n, m), dtype=torch.uint8, device=X.device) if out is None else out
    _lib.check(_lib.lib().rq_dev_encode_pq(_chk(out, torch.uint8, "codes"), _chk(X, torch.float32, "X"),
                                           _chk(Ccat, torch.float32, "C"), n, d, m, h, _stream()))
    return out


def encode_pq_filter_w(X, Ccat, m, h):
    """Test aid: (codes, W [n][m][h]) of the split encode kernel -- W = the bf16 matrix-core filter's values."""
    n, d = X.shape
    out = torch.empty((n, m), dtype=torch.uint8, device=X.device)
    W = torch.full((n, m, h), float("nan"), dtype=torch.float32, device=X.device)
    _lib.check(_lib.lib().rq_dev_encode_pq_filter_w(out.data_ptr(), W.data_ptr(), _chk(X, torch.float32, "X"),
                                                    _chk(Ccat, torch.float32, "C"), n, d, m, h, _stream()))
    return out, W


def polar_factor(G, method=0):
    """R = U V' of the d x d matrix G (src/OPQ.jl:112-113) -> (R as a torch [d][d] tensor with R[k][i] = Julia's R[k, i],
    ok, steps).  method 0 = Newton-Schulz, 1 = Jacobi SVD."""
    import ctypes
    d = G.shape[0]
    Rimg = torch.empty((d, d), dtype=torch.float32, device=G.device)
    st = (ctypes.c_int * 2)()
    torch.cuda.synchronize()
    _lib.check(_lib.lib().rq_dev_polar_factor(Rimg.data_ptr(), _chk(G, torch.float32, "G"), d, method,
                                              ctypes.cast(st, ctypes.c_void_p)))
    return Rimg.t().contiguous(), st[0] == 0, int(st[1])


def rotate_T(R, X, out=None):
    n, d = X.shape
    out = torch.empty_like(X) if out is None else out
    _lib.check(_lib.lib().rq_dev_rotate_T(_chk(out, torch.float32, "RX"), _chk(R, torch.float32, "R"),
                                          _chk(X, torch.float32, "X"), d, n, _stream()))
    return out


def encode_opq(X, R, Ccat, m, h, out=None):
    n, d = X.shape
    out = torch.empty((n, m), dtype=torch.uint8, device=X.device) if out is None else out
    _lib.check(_lib.lib().rq_dev_encode_opq(_chk(out, torch.uint8, "codes"), _chk(X, torch.float32, "X"),
                                            _chk(R, torch.float32, "R"), _chk(Ccat, torch.float32, "C"),
                                            n, d, m, h, _stream()))
    return out


def encode_rvq(Xr, C, out=None, want_counts=False):
    """quantize_rvq on resident tensors.  Xr (n, d) holds X and is OVERWRITTEN with the final residual;
    C (m, h, d).  Returns codes (n, m) uint8 [, counts (m, h) int32]."""
    n, d = Xr.shape
    m, h, _ = C.shape
    out = torch.empty((n, m), dtype=torch.uint8, device=Xr.device) if out is None else out
    counts = torch.zeros((m, h), dtype=torch.int32, device=Xr.device) if want_counts else None
    _lib.check(_lib.lib().rq_dev_encode_rvq(_chk(out, torch.uint8, "codes"), _chk(Xr, torch.float32, "Xr"),
                                            _chk(C, torch.float32, "C"), n, d, m, h,
                                            None if counts is None else counts.data_ptr(), _stream()))
    return (out, counts) if want_counts else out


def adc_lut(centers, queries):
    m, h, sub = centers.shape
    nq = queries.shape[0]
    lut = torch.empty((nq, m, 256), dtype=torch.float32, device=queries.device)
    _lib.check(_lib.lib().rq_dev_adc_lut(lut.data_ptr(), _chk(centers, torch.float32, "centers"),
                                         _chk(queries, torch.float32, "queries"), nq, m, sub, _stream()))
    return lut


class OrderedBase:
    """A resident code matrix in bank-aware row order (rq_dev_order_rows): `codes` [n][row_width] uint8 and `perm` [n]
    int32 (position -> original row; None for a base too small to order) are views into ONE device buffer."""

    def __init__(self, buf, codes, perm, n, m):
        self.buf, self.codes, self.perm, self.n, self.m = buf, codes, perm, n, m

    @property
    def shape(self):
        return (self.n, self.m)


def order_rows(codes):
    """Bank-aware order of a resident base: returns an OrderedBase for linscan(...).  The answer of a scan does not
    depend on it (ids are original row numbers); only the LDS bank conflicts of the table gathers do."""
    import ctypes as C
    n, m = codes.shape
    L = _lib.lib()
    nbytes = int(L.rq_order_bytes(n, m))
    mp = int(L.rq_scan_row_width(m))
    buf = torch.empty((nbytes + 15) // 16 * 2, dtype=torch.int64, device=codes.device)     # 16-byte aligned
    c_out, p_out = C.c_void_p(), C.c_void_p()
    _lib.check(L.rq_dev_order_rows(buf.data_ptr(), C.byref(c_out), C.byref(p_out), _chk(codes, torch.uint8, "codes"),
                                   n, m, _stream()))
    b8 = buf.view(torch.uint8)
    co = (c_out.value or buf.data_ptr()) - buf.data_ptr()
    oc = b8[co:co + n * mp].view(n, mp)
    perm = None
    if p_out.value:
        po = p_out.value - buf.data_ptr()
        perm = b8[po:po + 4 * n].view(torch.int32)
    return OrderedBase(buf, oc, perm, n, m)


def linscan(codes, centers, queries, k, id_offset=0, id_base=0, want_keys=False, out=None):
    """Scan one resident shard.  Returns (dists, ids) or packed sorted keys [nq][k] (int64 view of
    the uint64 keys) when want_keys.  `codes`: [n][m] uint8, or an OrderedBase (order_rows)."""
    nq, d = queries.shape
    dev = queries.device
    if isinstance(codes, OrderedBase):
        ob = codes
        n, m = ob.n, ob.m
        if want_keys:
            dists = ids = None
            keys = torch.empty((nq, k), dtype=torch.int64, device=dev) if out is None else out
        elif out is None:
            keys = None
            dists = torch.empty((nq, k), dtype=torch.float32, device=dev)
            ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
        else:
            keys = None
            dists, ids = out
        _lib.check(_lib.lib().rq_dev_linscan_ordered(
            None if dists is None else dists.data_ptr(), None if ids is None else ids.data_ptr(),
            None if keys is None else keys.data_ptr(), ob.codes.data_ptr(), None if ob.perm is None else ob.perm.data_ptr(),
            _chk(centers, torch.float32, "centers"), _chk(queries, torch.float32, "queries"), n, nq, m, d, k,
            id_offset, id_base, _stream()))
        return keys if want_keys else (dists, ids)
    n, m = codes.shape
    if want_keys:
        keys = torch.empty((nq, k), dtype=torch.int64, device=dev) if out is None else out
        _lib.check(_lib.lib().rq_dev_linscan(None, None, keys.data_ptr(), _chk(codes, torch.uint8, "codes"),
                                             _chk(centers, torch.float32, "centers"),
                                             _chk(queries, torch.float32, "queries"), n, nq, m, d, k,
                                             id_offset, id_base, _stream()))
        return keys
    if out is None:
        dists = torch.empty((nq, k), dtype=torch.float32, device=dev)
        ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
    else:
        dists, ids = out
    _lib.check(_lib.lib().rq_dev_linscan(dists.data_ptr(), ids.data_ptr(), None, _chk(codes, torch.uint8, "codes"),
                                         _chk(centers, torch.float32, "centers"),
                                         _chk(queries, torch.float32, "queries"), n, nq, m, d, k,
                                         id_offset, id_base, _stream()))
    return dists, ids


def linscan_aq(codes, codebooks, queries, k, dbnorms=None, id_offset=0, id_base=0, want_keys=False, out=None):
    """ADC scan for additive quantizers on resident tensors (rq_dev_linscan_aq): codebooks [m*256][d];
    dbnorms given -> LSQ tables -2<q,c> + per-row norm (src/Linscan.jl:118-157), else CQ tables |q-c|^2 (:160-193)."""
    n, m = codes.shape
    nq, d = queries.shape
    dev = codes.device
    mode = 1 if dbnorms is not None else 2
    nrm = None if dbnorms is None else _chk(dbnorms, torch.float32, "dbnorms")
    if want_keys:
        keys = torch.empty((nq, k), dtype=torch.int64, device=dev)
        _lib.check(_lib.lib().rq_dev_linscan_aq(None, None, keys.data_ptr(), _chk(codes, torch.uint8, "codes"),
                                                _chk(codebooks, torch.float32, "codebooks"),
                                                _chk(queries, torch.float32, "queries"), nrm, n, nq, m, d, k, mode,
                                                id_offset, id_base, _stream()))
        return keys
    if out is None:
        dists = torch.empty((nq, k), dtype=torch.float32, device=dev)
        ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
    else:
        dists, ids = out
    _lib.check(_lib.lib().rq_dev_linscan_aq(dists.data_ptr(), ids.data_ptr(), None, _chk(codes, torch.uint8, "codes"),
                                            _chk(codebooks, torch.float32, "codebooks"),
                                            _chk(queries, torch.float32, "queries"), nrm, n, nq, m, d, k, mode,
                                            id_offset, id_base, _stream()))
    return dists, ids


def merge_topk(keys_in, k, id_base=0, out=None):
    """keys_in [nq][P][k] int64 (uint64 bit patterns) -> (dists [nq][k], ids [nq][k])."""
    nq, P, kk = keys_in.shape
    assert kk == k
    dev = keys_in.device
    if out is None:
        dists = torch.empty((nq, k), dtype=torch.float32, device=dev)
        ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
    else:
        dists, ids = out
    _lib.check(_lib.lib().rq_dev_merge_topk(dists.data_ptr(), ids.data_ptr(), None,
                                            _chk(keys_in, torch.int64, "keys"), nq, P, k, id_base, _stream()))
    return dists, ids


def synth_codes(n, m, seed, row0=0, device="cuda"):
    codes = torch.empty((n, m), dtype=torch.uint8, device=device)
    _lib.check(_lib.lib().rq_dev_synth_codes(codes.data_ptr(), n, m, seed, row0, _stream()))
    return codes


# ---- training reductions --------------------------------------------------------------
def update_centers(Ccat, X, codes, m, h):
    """In place: every non-empty cluster's centre <- mean of its sub-vectors.  Returns counts [m][h]."""
    n, d = X.shape
    counts = torch.zeros((m, h), dtype=torch.int32, device=X.device)
    _lib.check(_lib.lib().rq_dev_update_centers(_chk(Ccat, torch.float32, "C"), counts.data_ptr(),
                                                _chk(X, torch.float32, "X"), _chk(codes, torch.uint8, "codes"),
                                                n, d, m, h, _stream()))
    return counts


def reconstruct(codes, Ccat, d, h, out=None):
    n, m = codes.shape
    out = torch.empty((n, d), dtype=torch.float32, device=codes.device) if out is None else out
    _lib.check(_lib.lib().rq_dev_reconstruct(_chk(out, torch.float32, "CB"), _chk(codes, torch.uint8, "codes"),
                                             _chk(Ccat, torch.float32, "C"), n, d, m, h, _stream()))
    return out


def qerror(X, CB):
    """mean_j |X_j - CB_j|^2 (python float)."""
    n, d = X.shape
    acc = torch.zeros((1,), dtype=torch.float64, device=X.device)
    _lib.check(_lib.lib().rq_dev_qerror(acc.data_ptr(), _chk(X, torch.float32, "X"), _chk(CB, torch.float32, "CB"),
                                        n, d, _stream()))
    return float(acc.item()) / n


def qerror_codes(X, codes, Ccat, h):
    """mean_j |X_j - CB_j|^2 with CB given as (codes, C): no n x d reconstruction."""
    n, d = X.shape
    m = codes.shape[1]
    acc = torch.zeros((1,), dtype=torch.float64, device=X.device)
    _lib.check(_lib.lib().rq_dev_qerror_codes(acc.data_ptr(), _chk(X, torch.float32, "X"), _chk(codes, torch.uint8, "codes"),
                                              _chk(Ccat, torch.float32, "C"), n, d, m, h, _stream()))
    return float(acc.item()) / n


def gram_codes(X, codes, Ccat, h):
    """G = X' CB with CB given as (codes, C)."""
    n, d = X.shape
    m = codes.shape[1]
    G = torch.empty((d, d), dtype=torch.float32, device=X.device)
    _lib.check(_lib.lib().rq_dev_gram_codes(G.data_ptr(), _chk(X, torch.float32, "X"), _chk(codes, torch.uint8, "codes"),
                                            _chk(Ccat, torch.float32, "C"), n, d, m, h, _stream()))
    return G


def gram(X, CB):
    """G = X' CB, [d][d] with G[a][b] = sum_j X[j][a] CB[j][b]."""
    n, d = X.shape
    G = torch.empty((d, d), dtype=torch.float32, device=X.device)
    _lib.check(_lib.lib().rq_dev_gram(G.data_ptr(), _chk(X, torch.float32, "X"), _chk(CB, torch.float32, "CB"),
                                      n, d, _stream()))
    return G
