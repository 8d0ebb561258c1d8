"""Pin the IS09 part of the CPU oracle (oracle/lld_oracle_is09.c: cAcf, cPitchACF,
cEnergy, cMZcr, cContourSmoother + delta chain) against golden LLD-level outputs
of the REAL reference binary (config/is09-13/IS09_emotion.conf, -lldhtkoutput)."""
import numpy as np
import pytest

KEYS = ["u2_16000", "u3_16000", "u10_16000", "u1_16000", "u0_16000", "u7_399", "u7_400", "u7_560",
        "u7_720", "u7_880", "u7_1040", "u4_48000"]

# columns: 0 RMS energy | 1..12 mfcc | 13 zcr | 14 voiceProb | 15 F0 | 16..31 their deltas
SCALE_GROUPS = [(slice(0, 1), "energy"), (slice(1, 13), "mfcc"), (slice(13, 14), "zcr"),
                (slice(14, 15), "voiceProb"), (slice(15, 16), "F0")]


@pytest.mark.parametrize("key", KEYS)
def test_is09_bit_exact_with_reference_fft(oracle, golden_is09, key):
    ref = golden_is09["out_" + key]
    if not oracle.use_reference_fft(True):
        pytest.skip("oracle/_ref/libref_dsp.so not built")
    try:
        out = oracle.is09_chain(golden_is09["pcm_" + key])
    finally:
        oracle.use_reference_fft(False)
    if ref.size == 0:
        assert out.shape[0] == 0
        return
    assert out.shape == ref.shape        # T+1 rows: the SMA's end-of-input frame survives (R13)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), f"max abs {np.abs(out - ref).max()}"


@pytest.mark.parametrize("key", KEYS)
def test_is09_builtin_fft_bit_exact(oracle, golden_is09, key):
    """Built-in transform (the oracle's own restatement of the rdft network, forward and inverse): the real binary's bits."""
    from tolerance import assert_bits_equal
    ref = golden_is09["out_" + key]
    oracle.use_reference_fft(False)
    out = oracle.is09_chain(golden_is09["pcm_" + key])
    if ref.size == 0:
        assert out.shape[0] == 0
        return
    assert_bits_equal(out, ref, key)