"""Pin the eGeMAPSv02 part of the CPU oracle (oracle/lld_oracle_gemaps.c: BASELINE config 5) against the REAL reference
binary: every internal level, the 25-column LLD level and the 88 functionals of config/egemaps/v02/eGeMAPSv02.conf, bit
for bit, on the golden file made by tests/golden/make_golden.py gen_egemaps (HTK taps of oracle/conf/egemaps_taps.conf)
and -- where oracle/_ref is built -- on fresh inputs through the live binary."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "egemaps_lld_synth.npz"))


KEYS = ["u2_16000", "u3_48000", "u10_16000", "u1_16000", "u0_16000", "u7_960", "u7_1120", "u7_1600", "u7_2720", "u4_9000",
        "u37_9000", "u2_8720", "u5_16000", "u11_160000", "u3_1600", "u3_1760", "u10_1280", "u2_1760", "u28_2240",
        "u4_1920", "u10_1440", "u3_1280", "u7_800"]
FRAME_LEVELS = ("loudness", "lspec", "flux", "mfcc", "energy2", "formants", "pitch", "jitter", "harm", "shs", "e60")
SMOOTHED = ("E", "F", "logf0", "loud", "NoZ", "NoNz", "specV", "specU")


def same(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))))


@pytest.mark.parametrize("key", KEYS)
def test_levels_lld_and_functionals_bit_exact_with_reference_fft(oracle, golden, key):
    if not oracle.use_reference_fft(True):
        pytest.skip("oracle/_ref/libref_dsp.so not built")
    try:
        pcm = golden["pcm_" + key]
        d = oracle.egemaps_levels(pcm)
        lld = oracle.egemaps_lld_chain(pcm)
        func = oracle.egemaps_func(pcm)
    finally:
        oracle.use_reference_fft(False)
    for k in FRAME_LEVELS:
        ref = golden[k + "_" + key]
        if ref.size == 0 and d[k].size == 0:
            continue
        assert same(d[k], ref), f"{k}: {d[k].shape} vs {ref.shape}"
    if d["T60"] >= 1:
        for k in SMOOTHED:
            assert same(d[k], golden[k + "_" + key]), k
    assert same(lld, golden["lld_" + key].reshape(-1, 25))
    assert same(func, golden["func_" + key].reshape(-1, 88))
    assert lld.shape[0] == (d["T60"] + 1 if d["T60"] >= 1 else 0)      # rows both smoothed levels hold


@pytest.mark.parametrize("key", ["u2_16000", "u10_16000", "u4_9000", "u7_1600"])
def test_builtin_fft_bit_exact(oracle, golden, key):
    """Without the hook the chain runs on the oracle's own restatement of the rdft network: the real binary's bits."""
    from tolerance import assert_bits_equal
    oracle.use_reference_fft(False)
    lld = oracle.egemaps_lld_chain(golden["pcm_" + key])
    assert_bits_equal(lld, golden["lld_" + key].reshape(lld.shape), key)


@pytest.mark.skipif(not __import__("oracle.lldo", fromlist=["x"]).have_ref(), reason="oracle/_ref not built")
def test_against_live_reference(oracle):
    """Fresh inputs (not in the golden file), incl. one without any 60 ms frame, through the real binary."""
    from opensmile_amd import synth
    if not oracle.use_reference_fft(True):
        pytest.skip("oracle/_ref/libref_dsp.so not built")
    try:
        for u, n in ((12, 24000), (13, 5000), (21, 1300), (14, 900)):
            pcm = synth.utterance(u, n)
            ref = oracle.run_reference_egemaps(pcm, levels=("F", "specV"))
            assert same(oracle.egemaps_lld_chain(pcm), ref["lld"].reshape(-1, 25)), (u, n)
            assert same(oracle.egemaps_func(pcm), ref["func"].reshape(-1, 88)), (u, n)
    finally:
        oracle.use_reference_fft(False)


def test_functional_specs_have_the_reference_counts(oracle):
    """Values per column of every instance add up to the 88 of the func level."""
    import ctypes as C
    L = oracle.lib()
    L.lldo_funcspec_count.restype = C.c_int
    counts = {i: L.lldo_funcspec_count(C.byref(oracle.egemaps_func_spec(i))) for i in
              ("F0", "Loudness", "MVZ", "MVV", "MU", "numPeaks", "segF0", "segF0pause", "leq")}
    assert counts == {"F0": 10, "Loudness": 10, "MVZ": 2, "MVV": 2, "MU": 1, "numPeaks": 1, "segF0": 3, "segF0pause": 2, "leq": 1}
    assert 10 + 10 + 5 * 2 + 23 * 2 + 5 * 1 + 1 + 3 + 2 + 1 == 88
