"""ctypes binding of librayuela_hip.so -- the ONLY compute backend of this package.

There is no CPU fallback: if the library is missing or cannot be loaded this module raises,
and every wrapper raises RayuelaHipError on a non-zero status (message from rq_last_error()).
The binding mirrors include/rayuela_hip.h one to one.
"""
import ctypes as C

import numpy as np
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


class RayuelaHipError(RuntimeError):
    pass


def lib_path():
    return os.environ.get("RAYUELA_HIP_LIB", os.path.join(_HERE, "librayuela_hip.so"))


_vp, _i32, _i64, _u32, _u64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_uint64

# name -> (restype, argtypes); kept in sync with include/rayuela_hip.h (tests/test_cabi.py checks it)
SIGNATURES = {
    "rq_version": (C.c_char_p, []),
    "rq_last_scan_kernel": (C.c_char_p, []),
    "rq_last_error": (C.c_char_p, []),
    "rq_device_count": (_i32, []),
    "rq_set_device": (_i32, [_i32]),
    "linscan_aqd_query": (None, [_vp, _vp, _vp, _vp, _vp, _i32, C.c_uint, _i32, _i32, _i32, _i32, _i32]),
    "linscan_aqd_query_extra_byte": (None, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32]),
    "linscan_aqd_cq_query_extra_byte": (None, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32]),
    "rq_linscan_lsq": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32]),
    "rq_linscan_cq": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32]),
    "rq_lsq_prepare": (_vp, [_vp, _vp, _vp, _i64, _i32, _i32, _i32]),
    "rq_lsq_search": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32]),
    "rq_lsq_release": (None, [_vp]),
    "rq_dev_linscan_aq": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _u32, _i32, _vp]),
    "rq_linscan_pq": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32]),
    "rq_linscan_opq": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32]),
    "rq_encode_pq": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32]),
    "rq_encode_opq": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32]),
    "rq_encode_pq_i16": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32]),
    "rq_encode_opq_i16": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32]),
    "rq_encode_rvq": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "rq_encode_rvq_i16": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "rq_train_rvq": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, C.c_uint64]),
    "rq_dev_encode_rvq": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "rq_dataset_upload": (_vp, [_vp, _i64, _i32]),
    "rq_dataset_encode": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32]),
    "rq_dataset_free": (None, [_vp]),
    "rq_rotate_T": (_i32, [_vp, _vp, _vp, _i32, _i64]),
    "rq_dev_encode_pq": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "rq_dev_encode_pq_filter_w": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "rq_last_encode_kernel": (C.c_char_p, []),
    "rq_last_encode_stats": (C.c_int, [C.c_void_p]),
    "rq_dev_rotate_T": (_i32, [_vp, _vp, _vp, _i32, _i64, _vp]),
    "rq_dev_encode_opq": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "rq_dev_adc_lut": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "rq_dev_linscan": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _u32, _i32, _vp]),
    "rq_scan_row_width": (_i32, [_i32]),
    "rq_order_bytes": (_i64, [_i64, _i32]),
    "rq_order_plan": (_i32, [_i64, _i32, _vp, _i32]),
    "rq_dev_order_rows": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "rq_dev_linscan_ordered": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _u32, _i32, _vp]),
    "rq_dev_merge_topk": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "rq_dev_synth_codes": (_i32, [_vp, _i64, _i32, _u64, _i64, _vp]),
    "rq_dev_update_centers": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "rq_dev_reconstruct": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "rq_dev_qerror": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "rq_dev_gram": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "rq_dev_gram_codes": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "rq_dev_qerror_codes": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "rq_kmpp_seeds": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _u64]),
    "rq_train_pq": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _u64]),
    "rq_train_opq": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _u64, _vp, _vp]),
    "rq_train_profile": (_i32, [_vp, _i32]),
    "rq_dev_polar_factor": (_i32, [_vp, _vp, _i32, _i32, _vp]),
    "rq_index_create": (_vp, [_i32, _i32, _vp]),
    "rq_index_create_sharded": (_vp, [_i32, _i32, _vp, _vp, _i32]),
    "rq_index_set_codes": (_i32, [_vp, _vp, _i64, _u32]),
    "rq_index_set_codes_synth": (_i32, [_vp, _i64, _u64, _u32]),
    "rq_index_search_opq": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32]),
    "rq_index_info": (_i32, [_vp, _vp, _i32]),
    "rq_release_workspaces": (_i32, []),
    "rq_host_alloc": (_vp, [C.c_size_t]),
    "rq_host_free": (None, [_vp]),
    "rq_index_search": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32]),
    "rq_index_destroy": (None, [_vp]),
    "rq_set_tuning": (_i32, [C.c_char_p, _i32]),
    "rq_scan_stats": (_i32, [_vp]),
    "rq_scan_orders_in_call": (_i32, [_i64, _i64, _i32]),
    "rq_scan_finish_stats": (_i32, [_vp]),
    "rq_scan_plan": (_i32, [_i64, _i64, _i32, _i32, _i32, _i32, _vp]),
    "rq_last_timing": (_i32, [_vp, _vp, _vp, _vp]),
}

_LIB = None


def lib():
    """Load the shared library (once).  Raises RayuelaHipError if it is not there."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.isfile(path):
            raise RayuelaHipError(
                "librayuela_hip.so not found at %s -- build it with `make -C rayuela.jl_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback" % path)
        try:
            # torch (when used for device memory / RCCL) bundles its own libamdhip64.so.7: importing it
            # first makes both sides share ONE HIP runtime, so device pointers are interchangeable.
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for the pure host-pointer API
            pass
        handle = C.CDLL(path)
        lenient = os.environ.get("RAYUELA_HIP_LENIENT") == "1"     # tools/ab_shard.py against an OLDER build: newer symbols may be absent
        for name, (res, args) in SIGNATURES.items():
            if lenient and not hasattr(handle, name):
                continue
            fn = getattr(handle, name)  # AttributeError here == ABI mismatch; let it surface
            fn.restype = res
            fn.argtypes = args
        _LIB = handle
    return _LIB


def check(status):
    if status != 0:
        msg = lib().rq_last_error().decode("utf-8", "replace")
        raise RayuelaHipError("librayuela_hip status %d: %s" % (status, msg))


def set_tuning(key, value):
    check(lib().rq_set_tuning(key.encode(), int(value)))


def scan_plan(n, nq, m, d, k, num_cu=256):
    """The planner's decision (host code only): dict of qg, groups, whole, slices, rows_per_slice, grid, cap, bigk, xcd."""
    out = (C.c_int64 * 8)()
    check(lib().rq_scan_plan(n, nq, m, d, k, num_cu, C.cast(out, C.c_void_p)))
    keys = ("qg", "groups", "whole", "slices", "rows_per_slice", "grid", "cap", "flags")
    p = dict(zip(keys, [int(x) for x in out]))
    p["bigk"], p["xcd"] = p["flags"] & 1, (p["flags"] >> 1) & 1
    return p


def scan_stats():
    out = (C.c_uint64 * 16)()
    check(lib().rq_scan_stats(C.cast(out, C.c_void_p)))
    names = ["lut", "sample", "stream", "cuts", "final_cut", "sort_write", "n_cuts", "n_fallbacks", "sample_rows", "sort_load", "sort_stages", "sort_out", "n_items", "n_items_filtered", "first_block_pushed", "first_block_rows"]
    st = dict(zip(names, [int(x) for x in out]))
    fin = (C.c_uint64 * 8)()
    check(lib().rq_scan_finish_stats(C.cast(fin, C.c_void_p)))
    st.update(zip(["bf_items", "bf_look_skips", "bf_select_sort"], [int(x) for x in fin[:3]]))
    return st


class _PinnedBlock:
    """A page-locked buffer of the library's pool, exposed through the array interface; returned to the pool when the
    last numpy view of it is collected."""

    def __init__(self, ptr, shape, dtype):
        self._ptr = ptr
        self.__array_interface__ = {"shape": tuple(shape), "typestr": np.dtype(dtype).str, "data": (ptr, False), "version": 3}

    def __del__(self):
        try:
            lib().rq_host_free(self._ptr)
        except Exception:     # noqa: BLE001 -- interpreter shutdown
            pass


# Results below this size are not worth a pool round trip (and small pinned allocations fragment the pool)
_PIN_MIN_BYTES = 4 << 20


def result_empty(shape, dtype):
    """An uninitialised result array: page-locked memory from the library's pool (rq_host_alloc) when the array is large,
    else numpy's own.  Page-locked results skip the first-touch page faults that otherwise bound the copy back
    (include/rayuela_hip.h); they are ordinary numpy arrays to the caller."""
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dtype.itemsize
    if nbytes >= _PIN_MIN_BYTES:
        ptr = lib().rq_host_alloc(nbytes)
        if ptr:
            return np.asarray(_PinnedBlock(ptr, shape, dtype))
    return np.empty(shape, dtype=dtype)


TRAIN_PHASES = ["h2d_ms", "init_ms", "qerror_ms", "gram_ms", "svd_ms", "rotate_ms", "update_centers_ms", "encode_ms",
                "reconstruct_ms", "converge_ms", "d2h_ms", "loop_ms", "iterations", "jacobi_sweeps", "ns_steps", "host_polar"]


def train_profile():
    """Phase clock of this thread's last train_pq / train_opq call (rq_train_profile)."""
    out = (C.c_double * 16)()
    check(lib().rq_train_profile(C.cast(out, C.c_void_p), 16))
    return dict(zip(TRAIN_PHASES, [float(x) for x in out]))


def last_timing():
    t = (C.c_double * 4)()
    p = [C.cast(C.byref(t, 8 * i), C.c_void_p) for i in range(4)]
    lib().rq_last_timing(*p)
    return dict(total_ms=t[0], h2d_ms=t[1], kernel_ms=t[2], d2h_ms=t[3])
