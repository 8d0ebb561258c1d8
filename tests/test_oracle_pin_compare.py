"""Pin the ComParE part of the CPU oracle (oracle/lld_oracle_compare.c: cSpectral with
ComParE_2016's options, cPlp auditory spectrum incl. RASTA, cVectorOperation ll1,
cEnergy / cMZcr on the 20 ms / 60 ms frames, MFCC 1-14, SMA + delta over levels of
different lengths) against golden LLD-level outputs of the REAL reference binary
(config/compare16/ComParE_2016.conf, -lldhtkoutput, columns of groups A and B)."""
import numpy as np
import pytest

KEYS = ["u2_16000", "u3_16000", "u10_16000", "u1_16000", "u0_16000", "u7_1440", "u7_1600", "u4_48000"]


@pytest.mark.parametrize("key", KEYS)
def test_compare_ab_bit_exact_with_reference_fft(oracle, golden_compare, key):
    ref = golden_compare["out_" + key]
    if not oracle.use_reference_fft(True):
        pytest.skip("oracle/_ref/libref_dsp.so not built")
    try:
        out = oracle.compare_ab_chain(golden_compare["pcm_" + key])
    finally:
        oracle.use_reference_fft(False)
    assert out.shape == ref.shape          # T60+1 rows x 118
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), f"max abs {np.abs(out - ref).max()}"


def compare_tolerances(out, ref, what=""):
    """Groups A + B [sma | delta] of ComParE_2016. Rounds 1-2 gated these per descriptor (5e-6 of the column scale, roll-off
    points allowed to move on 0.2 % of the cells) because the transform had its own butterfly order. Round 3: the oracle's
    built-in transform and the device's are the reference's rdft network, logf / expf are glibc's, the FLOAT_DMEM sums are
    sequential -- the columns are the reference's bits."""
    from tolerance import assert_bits_equal
    assert np.isfinite(out).all()
    assert_bits_equal(out, ref, what)


@pytest.mark.parametrize("key", KEYS)
def test_compare_ab_builtin_fft_bit_exact(oracle, golden_compare, key):
    """without the hook: the oracle's own restatement of the rdft network (lld_oracle_fft.c)"""
    oracle.use_reference_fft(False)
    out = oracle.compare_ab_chain(golden_compare["pcm_" + key])
    compare_tolerances(out, golden_compare["out_" + key], key)
