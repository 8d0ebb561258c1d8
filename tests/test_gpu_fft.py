"""The fused half-length transform of the wave-per-frame kernels (opensmile_amd/csrc/lld_fft.hpp) against the in-place radix-2
form it replaces: the same butterflies with the same table entries, only placed differently between stages -- so the two
must agree bit for bit on any input (that is what lets the front ends switch without touching a single parity gate)."""
import ctypes as C
import os

import numpy as np
import pytest


def _testlib():
    """tests/helpers/libsmilehip_testkernels.so (built by __graft_entry__.build() from tests/helpers/testkernels.hip): the
    building-block entry points live in a test helper library, not in the product's libsmilehip.so"""
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "libsmilehip_testkernels.so")
    if not os.path.exists(p):
        pytest.skip("tests/helpers/libsmilehip_testkernels.so not built (python __graft_entry__.py)")
    return C.CDLL(p)


@pytest.mark.gpu
@pytest.mark.parametrize("logm", [8, 9])
def test_fused_fft_is_bit_identical_to_radix2(logm):
    lib = _testlib()
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


@pytest.mark.gpu
def test_log_d_accuracy():
    """log_d (opensmile_amd/csrc/lld_device.hpp): the table + polynomial logarithm the frame kernels use for their per-bin
    logarithms, against numpy's long-double log. The reference computes these values with glibc's log (< 1 ulp) and rounds
    to float; an implementation within 1 ulp of double rounds to the same float except for near-ties (one in ~2^28)."""
    lib = _testlib()
    fn = lib.smilehip_debug_log_d
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.default_rng(3)
    parts = [np.exp(rng.uniform(-700, 700, 400000)), 1.0 + rng.uniform(-0.3, 0.4, 400000), 1.0 + rng.uniform(-1e-3, 1e-3, 200000),
             1.0 + np.exp(rng.uniform(-40, 0, 200000)), rng.uniform(0.0, 2.0, 200000).astype(np.float32).astype(np.float64) + 1.0,
             np.array([1.0, 2.0, 0.5, 0.6875, 1.375, np.nextafter(1.0, 0), np.nextafter(1.0, 2), 2.2250738585072014e-308, 1.7976931348623157e308])]
    x = np.concatenate(parts)
    y = np.empty_like(x)
    assert fn(x.ctypes.data, y.ctypes.data, len(x)) == 0
    ref = np.log(x.astype(np.longdouble))
    ulp = np.spacing(np.abs(ref.astype(np.float64))).astype(np.longdouble)
    err = np.abs(y.astype(np.longdouble) - ref) / ulp
    err[ref == 0] = np.abs(y[ref == 0])
    from tolerance import record
    record("log_d_ulp", max_ulp=float(err.max()), mean_ulp=float(err.mean()))
    # <= 1 ulp everywhere except where k ln2 + log c lands just above a binade boundary and the result just below it (the
    # rounding of that sum then counts double): <= 1.5 ulp there
    assert err.max() <= 1.5 and (err > 1.0).mean() <= 1e-4 and err.mean() <= 0.3, (float(err.max()), x[np.argmax(err)], float((err > 1.0).mean()))
    # as the kernels use it: rounded to float it equals the correctly rounded float logarithm
    f = y.astype(np.float32)
    assert (f != ref.astype(np.float32)).mean() <= 1e-6
    # special arguments take the library's path
    sp = np.array([0.0, -1.0, np.inf, np.nan, 5e-324])
    out = np.empty_like(sp)
    assert fn(sp.ctypes.data, out.ctypes.data, len(sp)) == 0
    assert out[0] == -np.inf and np.isnan(out[1]) and out[2] == np.inf and np.isnan(out[3]) and abs(out[4] - np.log(5e-324)) < 1e-9
