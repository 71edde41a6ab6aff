"""Device-resident index handle (rq_index_* of include/rayuela_hip.h): codes uploaded once, searched many
times -- on one MI355X or row-sharded over the GPUs of a node from this ONE process (the library gathers the
per-shard top-k lists over xGMI with RCCL and merges them on the first device).  Host arrays in, host arrays
out, like linscan_pq (src/Linscan.jl:5-26); ids are ONE-based by default like the reference's return value."""
import ctypes as C

import numpy as np

from . import _lib
from .Linscan import _centers, _codes_u8
from .utils import _as_f32

EXCHANGE = {0: "none", 1: "peer", 2: "rccl"}


class Index:
    def __init__(self, C_list, d, devices=None):
        """C_list: the m (256, d/m) codebooks; devices: None (current device) or a list of device ordinals,
        one shard per entry (a repeated ordinal = a logical shard on that device)."""
        m = len(C_list)
        self.m, self.d = m, d
        cen = _centers(C_list, m, d)
        lib = _lib.lib()
        if devices is None:
            self._h = lib.rq_index_create(m, d, cen.ctypes.data)
        else:
            devs = (C.c_int * len(devices))(*[int(x) for x in devices])
            self._h = lib.rq_index_create_sharded(m, d, cen.ctypes.data, C.cast(devs, C.c_void_p), len(devices))
        if not self._h:
            raise _lib.RayuelaHipError("rq_index_create: " + lib.rq_last_error().decode("utf-8", "replace"))
        self.n = 0

    def set_codes(self, B, id_offset=0):
        """B (n, m): uint8 zero-based codes, or another integer dtype holding ONE-based codes (src/Linscan.jl:28-37)."""
        Bu = _codes_u8(B)
        if Bu.shape[1] != self.m:
            raise ValueError("codes must have m=%d columns" % self.m)
        _lib.check(_lib.lib().rq_index_set_codes(self._h, Bu.ctypes.data, Bu.shape[0], id_offset))
        self.n = Bu.shape[0]
        return self

    def set_codes_synth(self, n, seed, id_offset=0):
        """SIFT1B-shape synthetic base generated on the devices (synth.random_codes gives the same bytes)."""
        _lib.check(_lib.lib().rq_index_set_codes_synth(self._h, int(n), int(seed), id_offset))
        self.n = int(n)
        return self

    def search(self, X, k=10000, R=None, id_base=1):
        X = _as_f32(X, "X")
        nq, d = X.shape
        if d != self.d:
            raise ValueError("queries must have d=%d columns" % self.d)
        dists = _lib.result_empty((nq, k), np.float32)
        idx = _lib.result_empty((nq, k), np.uint32)
        lib = _lib.lib()
        if R is None:
            _lib.check(lib.rq_index_search(self._h, dists.ctypes.data, idx.ctypes.data, X.ctypes.data, nq, k, id_base))
        else:
            R = _as_f32(R, "R")
            _lib.check(lib.rq_index_search_opq(self._h, dists.ctypes.data, idx.ctypes.data, X.ctypes.data,
                                               R.ctypes.data, nq, k, id_base))
        return dists, idx

    def info(self):
        out = (C.c_int64 * 68)()
        _lib.check(_lib.lib().rq_index_info(self._h, C.cast(out, C.c_void_p), 68))
        P = int(out[0])
        return {"shards": P, "devices": int(out[1]), "exchange": EXCHANGE.get(int(out[2]), "?"), "n": int(out[3]),
                "rows_per_shard": [int(out[4 + i]) for i in range(min(P, 64))]}

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().rq_index_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class Dataset:
    """A base set resident on the device (rq_dataset_*): uploaded once, encoded as often as needed."""

    def __init__(self, X):
        X = _as_f32(X, "X")
        self.n, self.d = X.shape
        self._h = _lib.lib().rq_dataset_upload(X.ctypes.data, self.n, self.d)
        if not self._h:
            raise _lib.RayuelaHipError("rq_dataset_upload: " + _lib.lib().rq_last_error().decode("utf-8", "replace"))

    def quantize(self, C_list, R=None, one_based=True):
        """quantize_pq(X, C) (R is None) or quantize_opq(X, R, C): (n, m) int16 one-based like the reference's
        return value, or uint8 zero-based (the scan's wire format) with one_based=False."""
        from .utils import cat_codebooks
        m = len(C_list)
        h = C_list[0].shape[0]
        Cc = np.ascontiguousarray(cat_codebooks(C_list), dtype=np.float32)
        Rp = None if R is None else _as_f32(R, "R")
        out = np.empty((self.n, m), dtype=np.int16 if one_based else np.uint8)
        _lib.check(_lib.lib().rq_dataset_encode(self._h, None if one_based else out.ctypes.data,
                                                out.ctypes.data if one_based else None,
                                                None if Rp is None else Rp.ctypes.data, Cc.ctypes.data, m, h))
        return out

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().rq_dataset_free(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
