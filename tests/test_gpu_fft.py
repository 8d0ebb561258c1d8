"""The fused half-length transform of the wave-per-frame kernels (opensmile_amd/csrc/lld_fft.hpp) against the in-place radix-2
form it replaces: the same butterflies with the same table entries, only placed differently between stages -- so the two
must agree bit for bit on any input (that is what lets the front ends switch without touching a single parity gate)."""
import ctypes as C

import numpy as np
import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("logm", [8, 9])
def test_fused_fft_is_bit_identical_to_radix2(logm):
    from opensmile_amd import capi
    lib = capi.load()
    fn = lib.smilehip_debug_fft_check
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    m, n = 1 << logm, 64
    rng = np.random.default_rng(7 + logm)
    x = rng.standard_normal((n, m, 2)).astype(np.float32)
    x[0] = 0.0
    x[0, 1, 0] = 1.0                      # an impulse, a constant, a few denormal-range and large values
    x[1] = 1.0
    x[2] *= 1e-30
    x[3] *= 1e20
    a = np.empty_like(x)
    b = np.empty_like(x)
    assert fn(logm, x.ctypes.data, a.ctypes.data, b.ctypes.data, n) == 0
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # and it is the transform: against numpy in double
    ref = np.fft.fft(x[4:, :, 0].astype(np.float64) + 1j * x[4:, :, 1].astype(np.float64), axis=1)
    got = b[4:, :, 0].astype(np.float64) + 1j * b[4:, :, 1]
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()
