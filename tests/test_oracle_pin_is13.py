"""config/is09-13/IS13_ComParE.conf -- ComParE_2016's graph with zeroPadSymmetric = 0 in both FFTs, cPitchJitter's
useBrokenJitterThresh = 1 and the functionals' ratio limiting / input normalisation switched off: the oracle's chain
restatements with the IS13 switch and the IS13 functionals specs against golden outputs of the real binary, bit for bit."""
import os

import numpy as np
import pytest

from test_oracle_pin_funcspec import ORDER, func_rows, inputs, pending

HERE = os.path.dirname(os.path.abspath(__file__))
KEYS = ["u3_48000", "u4_9000", "u7_1760", "u10_16000"]


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "is13_compare_synth.npz"))


@pytest.fixture()
def is13(oracle):
    oracle.compare_set_is13(True)
    yield oracle
    oracle.compare_set_is13(False)
    oracle.use_reference_fft(False)


@pytest.mark.parametrize("key", KEYS)
def test_is13_lld_level_bit_exact(is13, golden, key):
    is13.use_reference_fft(True)
    out = is13.compare_lld_chain(golden["pcm_" + key])
    ref = golden["lld130_" + key]
    assert out.shape == ref.shape
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("key", ["u4_9000", "u7_1760"])
def test_is13_functionals_bit_exact(is13, golden, key):
    T, P = pending(is13, golden["pcm_" + key])
    X = inputs(golden, key)
    f = golden["func_" + key]
    pos = 0
    for inst in ORDER:
        spec = is13.is13_func_spec(inst)
        per = len(is13.funcspec_names(spec))
        cols = X[inst].shape[1]
        ref = f[pos:pos + per * cols].reshape(cols, per)
        pos += per * cols
        out = is13.funcspec(np.ascontiguousarray(X[inst][:func_rows(inst, T, P)]), spec)
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), (key, inst)
    assert pos == 6373


def test_is13_differs_from_compare16(oracle, golden):
    """The switch matters: the 2016 chain on the same input is NOT the IS13 output (jitter columns, spectra)."""
    oracle.compare_set_is13(False)
    out = oracle.compare_lld_chain(golden["pcm_u3_48000"])
    ref = golden["lld130_u3_48000"]
    assert out.shape == ref.shape and not np.array_equal(out[:, 2:6], ref[:, 2:6])
