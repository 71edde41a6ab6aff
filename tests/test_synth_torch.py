"""The on-device (torch int64) generators of bench.py produce exactly the bytes of synth.py (numpy uint64)."""
import numpy as np
import pytest
import torch

import rayuela_jl_amd.synth as synth
import rayuela_jl_amd.synth_torch as st


@pytest.mark.parametrize("n,d,seed,nc,row0", [(3000, 128, 1234, 65536, 0), (777, 16, 4321, 1024, 12345), (100, 33, 99, 1000, 7)])
def test_sift_like_bit_identical(n, d, seed, nc, row0):
    a = synth.sift_like(n, d, seed=seed, ncentres=nc, row0=row0)
    b = st.sift_like(n, d, seed=seed, ncentres=nc, row0=row0, device="cpu", chunk=1000).numpy()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("n,d,seed,row0", [(5000, 96, 1234, 0), (300, 128, 4321, 999), (200, 200, 5, 3), (64, 960, 6, 0), (100, 7, 1, 0)])
def test_deep_like_bit_identical(n, d, seed, row0):
    a = synth.deep_like(n, d, seed=seed, row0=row0)
    b = st.deep_like(n, d, seed=seed, row0=row0, device="cpu", chunk=1000).numpy()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_random_codes_bit_identical():
    a = synth.random_codes(5000, 8, seed=1234, row0=17)
    b = st.random_codes(5000, 8, seed=1234, row0=17, device="cpu").numpy()
    assert np.array_equal(a, b)
    assert np.array_equal(st.splitmix64(torch.arange(10)).numpy().view(np.uint64), synth.splitmix64(np.arange(10)))


@pytest.mark.gpu
def test_device_generators_match_host():
    a = synth.sift_like(20000, 128, seed=1234, ncentres=65536)
    b = st.sift_like(20000, 128, seed=1234, ncentres=65536, device="cuda").cpu().numpy()
    assert np.array_equal(a, b)
    a = synth.deep_like(20000, 96, seed=1234)
    b = st.deep_like(20000, 96, seed=1234, device="cuda").cpu().numpy()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
