"""CPU: the C-ABI library loads and exports every symbol include/rayuela_hip.h declares
(no compute calls -- there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "rayuela_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(rq_[A-Za-z0-9_]+|linscan_aqd_[a-z_]*query[a-z_]*)\s*\(", src)
    return sorted(set(names))


def test_header_declares_the_reference_symbol():
    assert "linscan_aqd_query" in _declared_symbols()  # deps/src/linscan_aqd.cpp:105-114


def test_library_exports_every_declared_symbol(rq):
    handle = ctypes.CDLL(rq.lib_path())
    for name in _declared_symbols():
        assert hasattr(handle, name), "missing export: " + name


def test_python_binding_covers_the_header(rq):
    from rayuela_jl_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared_symbols()


def test_version_and_error_channel(rq):
    lib = rq.lib()
    assert b"gfx950" in lib.rq_version()
    assert isinstance(lib.rq_last_error(), bytes)


def test_no_cpu_fallback_without_a_device(rq):
    """On a box without an MI355X the product must fail loudly, not compute on the CPU."""
    import numpy as np
    if rq.lib().rq_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(rq.RayuelaHipError):
        rq.quantize_pq(np.zeros((4, 8), np.float32), [np.zeros((4, 4), np.float32)] * 2)
    with pytest.raises(rq.RayuelaHipError):
        rq.linscan_pq(np.zeros((4, 2), np.uint8), np.zeros((1, 4), np.float32),
                      [np.zeros((256, 2), np.float32)] * 2, 16, 1)


def test_missing_library_is_an_error(monkeypatch, rq):
    from rayuela_jl_amd import _lib
    monkeypatch.setenv("RAYUELA_HIP_LIB", "/nonexistent/librayuela_hip.so")
    monkeypatch.setattr(_lib, "_LIB", None)
    with pytest.raises(_lib.RayuelaHipError):
        _lib.lib()


def test_host_mirror_argument_checks(rq):
    import numpy as np
    # src/Linscan.jl:35: convert(Matrix{UInt8}, B .- 1) throws InexactError for codes outside 1..256
    with pytest.raises(OverflowError):
        rq.linscan_pq(np.array([[0, 1]], dtype=np.int16), np.zeros((1, 4), np.float32),
                      [np.zeros((256, 2), np.float32)] * 2, 16, 1)
    with pytest.raises(ValueError):  # Cint(d/m) InexactError, src/Linscan.jl:23
        rq.linscan_pq(np.zeros((4, 3), np.uint8), np.zeros((1, 4), np.float32),
                      [np.zeros((256, 1), np.float32)] * 3, 24, 1)
    with pytest.raises(TypeError):  # Float64 data never dispatches in the reference (src/PQ.jl:32)
        rq.quantize_pq(np.zeros((4, 8), np.float64), [np.zeros((4, 4), np.float32)] * 2)
    assert [list(p) for p in rq.splitarray(range(1, 11), 4)] == [[1, 2, 3], [4, 5, 6], [7, 8], [9, 10]]
