"""Host-side entry points added with the INTERSPEECH 2010 - 2012 components (no GPU needed): cSpecResample's geometry and tables
(smilehip_specresample_geometry / _tables) against the oracle's restatement of cSpecResample::setupNewNames + smileDsp_initIrdft,
the value counts and validation messages of the new functional families, and that the compute entry points refuse to run without
a device instead of falling back."""
import ctypes as C

import numpy as np
import pytest

from opensmile_amd import capi
from oracle import lldo


class _SpecRes(C.Structure):
    _fields_ = [("K", C.c_long), ("I", C.c_long), ("kMax", C.c_long), ("target_fs", C.c_double), ("costable", C.POINTER(C.c_float)),
                ("sintable", C.POINTER(C.c_float))]


@pytest.mark.parametrize("n_in,n_frame,rate,target", [(512, 400, 16000, 11000), (512, 320, 16000, 11000), (1024, 1024, 16000, 8000),
                                                      (256, 200, 8000, 11000), (2048, 1102, 44100, 11000), (512, 512, 16000, 16000),
                                                      (256, 256, 8000, 16000)])
def test_specresample_geometry_and_tables_equal_the_oracle(n_in, n_frame, rate, target):
    L = capi.load()
    fs_sec, last, bp = n_in / rate, n_frame / rate, 1.0 / rate
    n_out, k_max, nd = C.c_int64(0), C.c_int64(0), C.c_double(0.0)
    capi._check(L.smilehip_specresample_geometry(n_in, fs_sec, last, bp, float(target), C.byref(n_out), C.byref(k_max), C.byref(nd)))
    ol = lldo.lib()
    ol.lldo_specresample_init.argtypes = [C.POINTER(_SpecRes), C.c_long, C.c_double, C.c_double, C.c_double, C.c_double]
    ol.lldo_specresample_free.argtypes = [C.POINTER(_SpecRes)]
    r = _SpecRes()
    ol.lldo_specresample_init(C.byref(r), n_in, fs_sec, last, bp, float(target))
    try:
        assert (r.I, r.kMax) == (n_out.value, k_max.value)
        h = k_max.value // 2
        ct, st = np.zeros(h * n_out.value, np.float32), np.zeros(h * n_out.value, np.float32)
        capi._check(L.smilehip_specresample_tables(n_in, n_out.value, k_max.value, nd.value, ct.ctypes.data, st.ctypes.data))
        rc = np.ctypeslib.as_array(r.costable, shape=(h * n_out.value,))
        rs = np.ctypeslib.as_array(r.sintable, shape=(h * n_out.value,))
        assert np.array_equal(ct.view(np.uint32), rc.view(np.uint32)) and np.array_equal(st.view(np.uint32), rs.view(np.uint32))
    finally:
        ol.lldo_specresample_free(C.byref(r))


def test_new_functional_families_counts_and_validation():
    s = capi.FuncSpec()
    s.period = 0.01
    s.n_fam = 6
    for i, f in enumerate((9, 10, 11, 12, 13, 0)):          # Onset, Peaks, Crossings, DCT, Samples, Extremes
        s.fam[i] = f
    s.ons_mask, s.pko_mask, s.crs_mask, s.ext_mask = 0x1f, 0x1f, 0x7, 0x3
    s.dct_first, s.dct_last, s.n_samples = 1, 6, 5
    for i in range(5):
        s.sample_pos[i] = i / 4.0
    assert capi.funcspec_count(s) == 5 + 5 + 3 + 6 + 5 + 2
    s.ons_mask = 0x20
    with pytest.raises(capi.SmileHipError, match="Onset: unknown bits"):
        capi.funcspec_count(s)
    s.ons_mask, s.dct_last = 0x1f, 0
    with pytest.raises(capi.SmileHipError, match="DCT: coefficients"):
        capi.funcspec_count(s)
    s.dct_last, s.n_samples = 6, 9
    with pytest.raises(capi.SmileHipError, match="Samples: 1 .. 8"):
        capi.funcspec_count(s)
    s.n_samples = 2
    s.sample_pos[1] = 1.5
    with pytest.raises(capi.SmileHipError, match="samplepos"):
        capi.funcspec_count(s)


def test_new_operators_refuse_bad_arguments_without_a_device():
    """argument checks come before any device work: a null context is an error, never a CPU computation"""
    L = capi.load()
    buf = (C.c_float * 64)()
    for call in (lambda: L.smilehip_lsp_frames(None, buf, 8, 8, buf, 8, 1, None),
                 lambda: L.smilehip_intensity_frames(None, buf, 8, 8, 2, buf, 1, 1, None),
                 lambda: L.smilehip_vecop_frames(None, 2, 1.0, 0.0, buf, 8, 8, buf, 8, 1, None),
                 lambda: L.smilehip_lpc_acf_frames(None, buf, 32, 32, 8, buf, 8, 1, None),
                 lambda: L.smilehip_pitch_smoother_rows(None, 6, 0.7, 0, 1, 1, buf, 18, None, 1, 1, None, 0, buf, 1, None, None)):
        assert call() != 0
        assert "bad argument" in capi.load().smilehip_last_error().decode()
