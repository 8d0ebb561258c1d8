"""The oracle's Ooura-order real FFT (oracle/lld_oracle_fft.c) pinned bit for bit -- zero signs included -- against the REAL
rdft() compiled from /root/reference/src/dspcore/fftsg.c (oracle/_ref/libref_dsp.so), both directions, n = 64 ... 8192."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import lldo

REF = os.path.join(lldo.REF_DIR, "libref_dsp.so")


def _ref():
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/libref_dsp.so not built")
    return C.CDLL(REF)


def ref_rdft(ref, a, isgn):
    """rows of a (float32, n wide) through the reference's rdft with work areas sized as transformFft.cpp:201-208"""
    n = a.shape[1]
    out = np.ascontiguousarray(a, dtype=np.float32).copy()
    ip = np.zeros(3 + int(np.ceil(np.sqrt(n))) + 8, dtype=np.int32)
    w = np.zeros(n // 2 + 8, dtype=np.float32)
    fp = C.POINTER(C.c_float)
    for r in range(out.shape[0]):
        ref.rdft(C.c_int(n), C.c_int(isgn), out[r].ctypes.data_as(fp), ip.ctypes.data_as(C.POINTER(C.c_int)),
                 w.ctypes.data_as(fp))
    return out


def own_rdft(a, isgn):
    L = lldo.lib()
    n = a.shape[1]
    out = np.ascontiguousarray(a, dtype=np.float32).copy()
    fp = C.POINTER(C.c_float)
    for r in range(out.shape[0]):
        assert L.lldo_ooura_rdft(C.c_int(n), C.c_int(isgn), out[r].ctypes.data_as(fp)) == 0
    return out


def inputs(n, rows, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((rows, n)).astype(np.float32)
    x[0] = 0.0                                             # all zeros
    x[1] = -0.0                                            # all negative zeros
    x[2] = 0.0; x[2, 1] = 1.0                              # a delta (exact cancellations, signed zeros)
    x[3] = np.where(np.arange(n) % 160 < 80, 0.9, -0.9)    # square wave
    x[4] = np.round(x[4] * 4) / 4                          # few-bit values: exact cancellations
    x[5, n // 3:] = 0.0                                    # zero padded
    x[6] = np.where(rng.random(n) < 0.9, 0.0, x[6])        # sparse
    x[7] = np.where(rng.random(n) < 0.5, -0.0, 0.0)        # mixed zero signs
    return x


@pytest.mark.parametrize("n", [64, 128, 256, 512, 1024, 2048, 4096, 8192])
@pytest.mark.parametrize("isgn", [1, -1])
def test_bits_equal_reference(n, isgn):
    ref = _ref()
    x = inputs(n, 64 if n <= 1024 else 24, 100 + n)
    a = ref_rdft(ref, x, isgn)
    b = own_rdft(x, isgn)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), \
        f"n={n} isgn={isgn}: {np.count_nonzero(a.view(np.uint32) != b.view(np.uint32))} words differ"


def test_million_frames_512_1024():
    """10^6 random frames per size would take minutes in ctypes loops; 2 x 20 000 frames of int16-like
    audio here, the 10^6 run is tools/ooura_soak.py (result in profiles/r03_ooura_soak.json)."""
    ref = _ref()
    rng = np.random.default_rng(7)
    for n in (512, 1024):
        x = (rng.integers(-32768, 32767, size=(20000, n)).astype(np.float32) / np.float32(32767.0)).astype(np.float32)
        a = ref_rdft(ref, x, 1)
        b = own_rdft(x, 1)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_tables_equal_reference():
    ref = _ref()
    L = lldo.lib()
    fp = C.POINTER(C.c_float)
    for n in (64, 512, 1024, 4096):
        ip = np.zeros(64, dtype=np.int32)
        w = np.zeros(n // 2 + 8, dtype=np.float32)
        ref.makewt(C.c_int(n // 4), ip.ctypes.data_as(C.POINTER(C.c_int)), w.ctypes.data_as(fp))
        ref.makect(C.c_int(n // 4), ip.ctypes.data_as(C.POINTER(C.c_int)), w[n // 4:].ctypes.data_as(fp))
        wo = np.zeros(n // 4, dtype=np.float32)
        co = np.zeros(n // 4, dtype=np.float32)
        assert L.lldo_ooura_tables(C.c_int(n), wo.ctypes.data_as(fp), co.ctypes.data_as(fp)) == 0
        assert np.array_equal(w[:n // 4].view(np.uint32), wo.view(np.uint32))
        assert np.array_equal(w[n // 4:n // 2].view(np.uint32), co.view(np.uint32))
