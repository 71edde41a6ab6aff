"""HDF5 persistence of experiment results, the layout of demos/experiment_utils.jl:5-60.

The reference saves through HDF5.jl (`h5write(bpath, "$(trial)/C_$i", C[i])`, ...), i.e. through libhdf5.  This
module binds the same C library with ctypes -- no h5py needed: `libhdf5.so` is looked up in
$RAYUELA_HDF5_LIB, the loader path and the usual prefixes (/opt/conda/lib, /usr/lib/x86_64-linux-gnu/hdf5/serial).
If none is found every function raises Hdf5Unavailable.

Array convention = the one of the whole package: a numpy array is the C view (memory image) of the Julia array.
HDF5.jl stores a Julia `d x n` matrix with the dataspace dims reversed, (n, d) -- exactly the shape of our view --
so files written here are read by `h5read` in Julia as the arrays the reference would have written, and files
written by the reference load here as the arrays the rest of the package uses:
    C_i        (h, sub_i) float32     == Julia sub_i x h
    B, B_base  (n, m) uint8 ZERO-based == Julia m x n  `convert(Matrix{UInt8}, B .- 1)`   (:10,17)
    R          (d, d)  float32        memory image of Julia's R
    train_error scalar, recall (k,) float64
"""
import ctypes as C
import ctypes.util
import glob
import os

import numpy as np


class Hdf5Unavailable(RuntimeError):
    pass


_H = None
_T = {}

_hid = C.c_int64           # hid_t is int64_t since HDF5 1.10
H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC, H5F_ACC_EXCL = 0, 1, 2, 4
H5P_DEFAULT, H5S_ALL, H5S_SCALAR = 0, 0, 0
H5T_INTEGER, H5T_FLOAT = 0, 1
H5T_SGN_NONE, H5T_SGN_2 = 0, 1

_NATIVE = {np.dtype(np.float32): "H5T_NATIVE_FLOAT_g", np.dtype(np.float64): "H5T_NATIVE_DOUBLE_g",
           np.dtype(np.uint8): "H5T_NATIVE_UINT8_g", np.dtype(np.int8): "H5T_NATIVE_INT8_g",
           np.dtype(np.uint16): "H5T_NATIVE_UINT16_g", np.dtype(np.int16): "H5T_NATIVE_INT16_g",
           np.dtype(np.uint32): "H5T_NATIVE_UINT32_g", np.dtype(np.int32): "H5T_NATIVE_INT32_g",
           np.dtype(np.uint64): "H5T_NATIVE_UINT64_g", np.dtype(np.int64): "H5T_NATIVE_INT64_g"}


def _candidates():
    env = os.environ.get("RAYUELA_HDF5_LIB")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*",
                "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/local/lib/libhdf5.so*", "/usr/lib64/libhdf5.so*"):
        for p in sorted(glob.glob(pat)):
            if "_hl" not in p and "_cpp" not in p and "_fortran" not in p:
                yield p


def lib():
    """dlopen libhdf5 once and declare the handful of entry points used."""
    global _H
    if _H is not None:
        return _H
    last = None
    for path in _candidates():
        try:
            h = C.CDLL(path)
            h.H5open()
            break
        except OSError as e:
            last = e
    else:
        raise Hdf5Unavailable("libhdf5 not found (set RAYUELA_HDF5_LIB); last error: %s" % last)
    sig = {
        "H5Fcreate": (_hid, [C.c_char_p, C.c_uint, _hid, _hid]), "H5Fopen": (_hid, [C.c_char_p, C.c_uint, _hid]),
        "H5Fclose": (C.c_int, [_hid]), "H5Lexists": (C.c_int, [_hid, C.c_char_p, _hid]),
        "H5Pcreate": (_hid, [_hid]), "H5Pclose": (C.c_int, [_hid]),
        "H5Pset_create_intermediate_group": (C.c_int, [_hid, C.c_uint]),
        "H5Screate_simple": (_hid, [C.c_int, C.c_void_p, C.c_void_p]), "H5Screate": (_hid, [C.c_int]),
        "H5Sclose": (C.c_int, [_hid]), "H5Sget_simple_extent_ndims": (C.c_int, [_hid]),
        "H5Sget_simple_extent_dims": (C.c_int, [_hid, C.c_void_p, C.c_void_p]),
        "H5Dcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid, _hid]), "H5Dopen2": (_hid, [_hid, C.c_char_p, _hid]),
        "H5Dwrite": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
        "H5Dread": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
        "H5Dget_space": (_hid, [_hid]), "H5Dget_type": (_hid, [_hid]), "H5Dclose": (C.c_int, [_hid]),
        "H5Tget_class": (C.c_int, [_hid]), "H5Tget_size": (C.c_size_t, [_hid]), "H5Tget_sign": (C.c_int, [_hid]),
        "H5Tclose": (C.c_int, [_hid]), "H5Eset_auto2": (C.c_int, [_hid, C.c_void_p, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, args
    h.H5Eset_auto2(0, None, None)            # errors are reported through return codes -> Python exceptions
    for dt, sym in _NATIVE.items():
        _T[dt] = _hid.in_dll(h, sym).value
    _T["lcpl_class"] = _hid.in_dll(h, "H5P_CLS_LINK_CREATE_ID_g").value
    _H = h
    return h


def available():
    try:
        lib()
        return True
    except Hdf5Unavailable:
        return False


def _chk(v, what):
    if v < 0:
        raise IOError("libhdf5: %s failed" % what)
    return v


def h5write(path, name, data):
    """HDF5.jl's h5write(path, name, data): create the file if needed, create intermediate groups, write ONE new
    dataset (an existing name is an error, as in the reference)."""
    h = lib()
    a = np.asarray(data)
    if a.dtype == np.bool_:
        a = a.astype(np.uint8)
    if a.dtype not in _NATIVE:
        raise TypeError("h5write: unsupported dtype %s" % a.dtype)
    a = np.ascontiguousarray(a) if a.ndim else a.copy()      # (ascontiguousarray would turn a scalar into 1-d)
    bpath = os.fsencode(path)
    f = h.H5Fopen(bpath, H5F_ACC_RDWR, H5P_DEFAULT) if os.path.isfile(path) else -1
    if f < 0:
        f = _chk(h.H5Fcreate(bpath, H5F_ACC_EXCL if not os.path.exists(path) else H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), "H5Fcreate")
    space = lcpl = dset = -1
    try:
        if h.H5Lexists(f, name.split("/")[0].encode(), H5P_DEFAULT) > 0:
            # walk down: every prefix must exist before H5Lexists may be asked about the full path
            parts, ok = name.split("/"), True
            for i in range(1, len(parts) + 1):
                if h.H5Lexists(f, "/".join(parts[:i]).encode(), H5P_DEFAULT) <= 0:
                    ok = False
                    break
            if ok:
                raise IOError("h5write: %s already exists in %s" % (name, path))
        if a.ndim == 0:
            space = _chk(h.H5Screate(H5S_SCALAR), "H5Screate")
        else:
            dims = (C.c_uint64 * a.ndim)(*a.shape)
            space = _chk(h.H5Screate_simple(a.ndim, dims, None), "H5Screate_simple")
        lcpl = _chk(h.H5Pcreate(_T["lcpl_class"]), "H5Pcreate")
        _chk(h.H5Pset_create_intermediate_group(lcpl, 1), "H5Pset_create_intermediate_group")
        dset = _chk(h.H5Dcreate2(f, name.encode(), _T[a.dtype], space, lcpl, H5P_DEFAULT, H5P_DEFAULT), "H5Dcreate2(%s)" % name)
        _chk(h.H5Dwrite(dset, _T[a.dtype], H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data), "H5Dwrite")
    finally:
        if dset >= 0:
            h.H5Dclose(dset)
        if lcpl >= 0:
            h.H5Pclose(lcpl)
        if space >= 0:
            h.H5Sclose(space)
        h.H5Fclose(f)


def h5read(path, name):
    """HDF5.jl's h5read(path, name) -> numpy array in the C view (shape = the dataspace dims)."""
    h = lib()
    f = _chk(h.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, H5P_DEFAULT), "H5Fopen(%s)" % path)
    dset = space = ftype = -1
    try:
        dset = _chk(h.H5Dopen2(f, name.encode(), H5P_DEFAULT), "H5Dopen2(%s)" % name)
        space = _chk(h.H5Dget_space(dset), "H5Dget_space")
        nd = _chk(h.H5Sget_simple_extent_ndims(space), "ndims")
        dims = (C.c_uint64 * max(nd, 1))()
        if nd:
            _chk(h.H5Sget_simple_extent_dims(space, dims, None), "dims")
        ftype = _chk(h.H5Dget_type(dset), "H5Dget_type")
        cls, size = h.H5Tget_class(ftype), h.H5Tget_size(ftype)
        if cls == H5T_FLOAT:
            dt = {4: np.float32, 8: np.float64}[size]
        elif cls == H5T_INTEGER:
            signed = h.H5Tget_sign(ftype) == H5T_SGN_2
            dt = {(1, False): np.uint8, (1, True): np.int8, (2, False): np.uint16, (2, True): np.int16,
                  (4, False): np.uint32, (4, True): np.int32, (8, False): np.uint64, (8, True): np.int64}[(size, signed)]
        else:
            raise TypeError("h5read: dataset %s has an unsupported type class %d" % (name, cls))
        out = np.empty(tuple(int(dims[i]) for i in range(nd)), dtype=dt)
        _chk(h.H5Dread(dset, _T[np.dtype(dt)], H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data), "H5Dread")
        return out
    finally:
        if ftype >= 0:
            h.H5Tclose(ftype)
        if space >= 0:
            h.H5Sclose(space)
        if dset >= 0:
            h.H5Dclose(dset)
        h.H5Fclose(f)


def _codes_zero_based_u8(B):
    """convert(Matrix{UInt8}, B .- 1) (:10): B is the package's one-based Int16 (n, m); uint8 input is taken as
    already zero-based (the scan's wire format)."""
    B = np.asarray(B)
    if B.dtype == np.uint8:
        return B
    Bm1 = B.astype(np.int64) - 1
    if Bm1.min(initial=0) < 0 or Bm1.max(initial=0) > 255:
        raise OverflowError("InexactError: a one-based code outside 1..256 does not fit UInt8")
    return Bm1.astype(np.uint8)


# ---- demos/experiment_utils.jl:5-46 ---------------------------------------------------------------------------
def save_results_pq_query_base(bpath, trial, C, B, train_error, recall):
    for i, Ci in enumerate(C):
        h5write(bpath, "%d/C_%d" % (trial, i + 1), np.asarray(Ci, dtype=np.float32))
    h5write(bpath, "%d/B" % trial, _codes_zero_based_u8(B))
    h5write(bpath, "%d/train_error" % trial, np.asarray(train_error))
    h5write(bpath, "%d/recall" % trial, np.asarray(recall))


def save_results_pq(bpath, trial, C, B, train_error, B_base, recall):
    h5write(bpath, "%d/B_base" % trial, _codes_zero_based_u8(B_base))
    save_results_pq_query_base(bpath, trial, C, B, train_error, recall)


def save_results_opq_query_base(bpath, trial, C, B, R, train_error, recall):
    h5write(bpath, "%d/R" % trial, np.asarray(R, dtype=np.float32))
    save_results_pq_query_base(bpath, trial, C, B, train_error, recall)


def save_results_opq(bpath, trial, C, B, R, train_error, B_base, recall):
    h5write(bpath, "%d/B_base" % trial, _codes_zero_based_u8(B_base))
    save_results_opq_query_base(bpath, trial, C, B, R, train_error, recall)


def save_results_lsq_query_base(bpath, trial, C, B, R, train_error, opq_error, recall):
    h5write(bpath, "%d/opq_base" % trial, np.asarray(opq_error))
    save_results_opq_query_base(bpath, trial, C, B, R, train_error, recall)


def save_results_lsq(bpath, trial, C, B, R, train_error, opq_error, B_base, recall):
    h5write(bpath, "%d/B_base" % trial, _codes_zero_based_u8(B_base))
    save_results_lsq_query_base(bpath, trial, C, B, R, train_error, opq_error, recall)


# ---- demos/experiment_utils.jl:48-60 (+ the PQ / OPQ loaders the demos would need to resume) --------------------
def _load_codebooks(fname, m, trial):
    return [h5read(fname, "%d/C_%d" % (trial, i + 1)) for i in range(m)]


def _load_B(fname, trial, name="B"):
    return h5read(fname, "%d/%s" % (trial, name)).astype(np.int16) + 1      # :49  convert(Matrix{Int16}, B); B .+= 1


def load_chainq(fname, m, trial):
    """-> C, B (one-based Int16), R, error          (demos/experiment_utils.jl:48-55)"""
    return _load_codebooks(fname, m, trial), _load_B(fname, trial), h5read(fname, "%d/R" % trial), \
        h5read(fname, "%d/train_error" % trial)


def load_rvq(fname, m, trial):
    """-> C, B (one-based Int16), error             (demos/experiment_utils.jl:57-63)"""
    return _load_codebooks(fname, m, trial), _load_B(fname, trial), h5read(fname, "%d/train_error" % trial)


def load_pq(fname, m, trial):
    return load_rvq(fname, m, trial)


def load_opq(fname, m, trial):
    return load_chainq(fname, m, trial)
