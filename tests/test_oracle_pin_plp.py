"""Pin the PLP-CC part of the CPU oracle (cPlp with IDFT, Durbin recursion, LP -> cepstra, lifter:
oracle/lld_oracle_compare.c) against golden outputs of the REAL reference binary
(config/plp/PLP_0_D_A.conf): bit-exact with the reference's own rdft plugged in."""
import numpy as np
import pytest

KEYS = ["u2_16000", "u3_16000", "u10_16000", "u1_16000", "u0_16000", "u7_399", "u7_400", "u7_560", "u7_1000", "u5_160000"]


@pytest.mark.parametrize("key", KEYS)
def test_plp_bit_exact_with_reference_fft(oracle, golden_plp, key):
    ref = golden_plp["out_" + key]
    if not oracle.use_reference_fft(True):
        pytest.skip("oracle/_ref/libref_dsp.so not built")
    try:
        out = oracle.plp_chain(golden_plp["pcm_" + key])
    finally:
        oracle.use_reference_fft(False)
    assert out.shape == ref.shape
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("key", KEYS)
def test_plp_builtin_fft_bit_exact(oracle, golden_plp, key):
    oracle.use_reference_fft(False)
    ref = golden_plp["out_" + key]
    out = oracle.plp_chain(golden_plp["pcm_" + key])
    from tolerance import assert_bits_equal
    assert_bits_equal(out, ref, key)      # round 3: the oracle's built-in transform is the reference's rdft network
