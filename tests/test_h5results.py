"""HDF5 persistence (demos/experiment_utils.jl:5-60) through libhdf5 bound with ctypes: round trips, the reference's
dataset names and conventions (zero-based UInt8 codes on disk, one-based Int16 after loading, Julia's reversed
dataspace dims), and an independent check of the written file with libhdf5's own h5dump when it is installed."""
import os
import shutil
import subprocess

import numpy as np
import pytest

h5 = pytest.importorskip("rayuela_jl_amd.h5results")
if not h5.available():
    pytest.skip("libhdf5 not found", allow_module_level=True)


def test_h5write_h5read_round_trip(tmp_path):
    path = str(tmp_path / "a.h5")
    rng = np.random.default_rng(0)
    arrays = {"f32": rng.standard_normal((5, 3)).astype(np.float32), "f64": rng.standard_normal(7),
              "u8": rng.integers(0, 256, (4, 8)).astype(np.uint8), "i16": rng.integers(-300, 300, (2, 3, 4)).astype(np.int16),
              "u32": rng.integers(0, 2 ** 32, 9, dtype=np.uint64).astype(np.uint32), "scalar": np.float32(3.25)}
    for k, v in arrays.items():
        h5.h5write(path, "grp/sub/" + k, v)
    for k, v in arrays.items():
        got = h5.h5read(path, "grp/sub/" + k)
        assert got.dtype == np.asarray(v).dtype and got.shape == np.asarray(v).shape and np.array_equal(got, v)
    with pytest.raises(IOError):
        h5.h5write(path, "grp/sub/f32", arrays["f32"])            # HDF5.jl refuses to overwrite, too
    with pytest.raises(IOError):
        h5.h5read(path, "grp/nope")


def test_experiment_results_layout(tmp_path):
    """save_results_opq / load_chainq and save_results_pq / load_rvq: names, types and index bases of the reference."""
    path = str(tmp_path / "results.h5")
    rng = np.random.default_rng(1)
    m, h, d, n, nb, k = 4, 16, 12, 50, 70, 10
    C = [rng.standard_normal((h, d // m)).astype(np.float32) for _ in range(m)]
    B = rng.integers(1, h + 1, (n, m)).astype(np.int16)             # one-based, as quantize_pq returns it
    Bb = rng.integers(1, h + 1, (nb, m)).astype(np.int16)
    R = rng.standard_normal((d, d)).astype(np.float32)
    recall = np.linspace(0.1, 1.0, k)
    h5.save_results_opq(path, 3, C, B, R, np.float32(12.5), Bb, recall)
    # on disk: zero-based UInt8 codes (experiment_utils.jl:10,17), dataspace dims == our C-view shapes
    raw = h5.h5read(path, "3/B")
    assert raw.dtype == np.uint8 and raw.shape == (n, m) and np.array_equal(raw, (B - 1).astype(np.uint8))
    assert np.array_equal(h5.h5read(path, "3/B_base"), (Bb - 1).astype(np.uint8))
    assert h5.h5read(path, "3/C_2").shape == (h, d // m)
    C2, B2, R2, err = h5.load_chainq(path, m, 3)
    assert all(np.array_equal(a, b) for a, b in zip(C, C2)) and np.array_equal(B2, B) and B2.dtype == np.int16
    assert np.array_equal(R2, R) and float(err) == 12.5
    assert np.array_equal(h5.h5read(path, "3/recall"), recall)
    # a second trial goes into the same file
    h5.save_results_pq(path, 4, C, B, 1.5, Bb, recall)
    C3, B3, err3 = h5.load_rvq(path, m, 4)
    assert np.array_equal(B3, B) and float(err3) == 1.5
    with pytest.raises(OverflowError):
        h5.save_results_pq(path, 5, C, np.zeros((3, m), dtype=np.int16), 0.0, Bb, recall)   # code 0 is not one-based


def test_file_is_valid_hdf5_for_other_readers(tmp_path):
    dump = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.isfile("/opt/conda/bin/h5dump") else None)
    if not dump:
        pytest.skip("no h5dump")
    path = str(tmp_path / "v.h5")
    h5.h5write(path, "1/B", np.arange(6, dtype=np.uint8).reshape(3, 2))
    h5.h5write(path, "1/train_error", np.float32(0.5))
    out = subprocess.run([dump, path], capture_output=True, text=True)
    assert out.returncode == 0
    assert 'GROUP "1"' in out.stdout and 'DATASET "B"' in out.stdout and "H5T_STD_U8LE" in out.stdout
    assert "( 3, 2 )" in out.stdout and "0.5" in out.stdout
