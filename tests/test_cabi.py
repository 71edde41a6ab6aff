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


def test_scan_planner_decisions():
    """The planner is pure host code: its work-item decomposition can be checked without a GPU."""
    from rayuela_jl_amd import _lib
    # headline shape: 1250 groups >= 512 resident workgroups, remainder 226 > num_cu/2 -> everything whole
    p = _lib.scan_plan(1_000_000, 10_000, 8, 128, 1000)
    assert p["qg"] == 8 and p["groups"] == 1250 and p["whole"] == 1250 and p["slices"] == 1 and p["grid"] == 512
    assert p["bigk"] == 0 and p["cap"] >= 1000 + 4000
    # 5000 queries: 625 groups = 512 whole + 113 (<= 128) cut in two
    p = _lib.scan_plan(1_000_000, 5_000, 8, 128, 100)
    assert (p["whole"], p["slices"]) == (512, 2) and p["rows_per_slice"] * 2 >= 1_000_000
    # small batch: slices = resident workgroups / groups, never shorter than max(16384, 32 k) rows
    p = _lib.scan_plan(1_000_000, 1_000, 8, 128, 1000)
    assert (p["whole"], p["slices"]) == (0, 4)
    p = _lib.scan_plan(1_000_000, 8, 8, 128, 10)
    assert p["whole"] == 0 and p["slices"] == 41 and p["rows_per_slice"] == 24576   # >= 16384, whole 8192-row blocks
    p = _lib.scan_plan(1_000_000, 8, 8, 128, 10_000)
    assert p["slices"] == 3 and p["bigk"] == 1           # 32 k rows per slice at least; sample-sort finish
    # m = 16 groups 8 queries too, in ONE 1024-thread workgroup per CU (96 KiB of f32 tables + 32 KiB of byte tables);
    # m = 32 groups 4; padded widths plan like the next tiled width
    p = _lib.scan_plan(1_000_000, 10_000, 16, 96, 1000)
    assert p["qg"] == 8 and p["grid"] == 256 and p["whole"] == 1250
    assert _lib.scan_plan(1_000_000, 1_000, 32, 128, 100)["qg"] == 4
    assert _lib.scan_plan(1_000_000, 1_000, 12, 96, 100) == _lib.scan_plan(1_000_000, 1_000, 16, 96, 100)
    # SIFT1B shard: 1.25e8 rows, 1024 queries -> 128 groups over 32 MB row windows (round 2: 4 slices)
    p = _lib.scan_plan(125_000_000, 1024, 8, 128, 100)
    assert (p["groups"], p["slices"], p["xcd"]) == (128, 30, 1)


def test_xcd_window_plan_is_the_default_for_big_bases():
    from rayuela_jl_amd import _lib
    p = _lib.scan_plan(1_000_000_000, 1024, 8, 128, 100)              # BASELINE config 5's workload on one GPU
    assert p["xcd"] == 1 and p["whole"] == 0 and p["slices"] >= 128 and p["rows_per_slice"] * 8 <= 40 << 20, p
    p = _lib.scan_plan(125_000_000, 1024, 8, 128, 100)                # ... and one GPU's shard of it on 8 GPUs
    assert p["xcd"] == 1 and 16 <= p["slices"] <= 64, p
    assert _lib.scan_plan(1_000_000, 10_000, 8, 128, 1000)["xcd"] == 0   # SIFT1M shape: whole-base items


def test_order_plan_host_logic():
    """rq_order_plan (no device): the bank-aware row order spends log2(n / 32) key bits, 3 per leading code byte -- a
    32-value window per byte = one LDS bank column each (csrc/rq_order.hip)."""
    import ctypes as C
    from rayuela_jl_amd import _lib
    L = _lib.lib()

    def plan(n, m):
        out = (C.c_int * 14)()
        assert L.rq_order_plan(n, m, C.cast(out, C.c_void_p), 14) == 0
        return list(out)

    # round 6: where the greedy balance runs (8- and 16-byte rows, two-level sort) the key gives up one table's bits and the rows
    # of a sort bucket are dealt over its lane groups by the uncovered tables: 12 bits + 4 tables at SIFT1M shape
    p = plan(1_000_000, 8)
    assert p[:8] == [3, 3, 3, 3, 0, 0, 0, 0] and p[8] == 12 and p[9] == 32 and p[11] == 8 and p[12] == 4 and p[13] == 16
    from rayuela_jl_amd import set_tuning
    set_tuning("ORDER_GREEDY", 0)
    try:
        p = plan(1_000_000, 8)
        assert p[:8] == [3, 3, 3, 3, 3, 0, 0, 0] and p[8] == 15 and p[12] == 0
        assert plan(1_000_000, 16)[:8] == [3, 3, 3, 3, 3, 0, 0, 0]     # m = 16: 5 of 16 tables
    finally:
        set_tuning("ORDER_GREEDY", 1)
    p = plan(1_000_000, 16)
    assert p[:8] == [3, 3, 3, 3, 0, 0, 0, 0] and p[12] == 12 and 8 <= p[13] <= 16
    assert plan(1_000_000, 4)[12] == 0                             # 4-byte rows: the key covers every table already
    p = plan(125_000_000, 8)                       # the per-GPU shard of BASELINE config 5: 22 bits
    assert p[:8] == [3, 3, 3, 3, 3, 3, 3, 1] and p[8] == 22
    assert plan(1_000_000_000, 8)[:9] == [3] * 8 + [24]            # every table conflict-free from 2^29 rows on
    assert plan(500, 8)[8] == 0                                    # tiny base: no ordering
    p = plan(200_000, 5)                                           # padded to 8 bytes; 15/16 of the rows are sorted
    assert p[11] == 8 and p[8] in (12, 13) and p[:4] == [3, 3, 3, 3]
    p = plan(65_536, 2)                                            # narrow rows: both bytes first get a window, then more bits
    assert p[8] == 11 and p[0] + p[1] == 11 and p[2:8] == [0] * 6
    out = (C.c_int * 12)()
    assert L.rq_order_plan(1000, 65, C.cast(out, C.c_void_p), 12) != 0          # m > 64
