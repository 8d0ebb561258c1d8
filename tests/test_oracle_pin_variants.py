"""Pin the oracle's restatement of the other configs of config/mfcc and config/plp (MFCC12_E_D_A, MFCC12_0_D_A_Z,
MFCC12_E_D_A_Z, PLP_E_D_A, PLP_0_D_A_Z, PLP_E_D_A_Z: cEnergy's HTK log energy as an extra static column, cFullinputMean
on the cepstra, symmetric zero padding in the _Z files) against golden outputs of the REAL reference binary."""
import numpy as np
import pytest

NAMES = ["MFCC12_E_D_A", "MFCC12_0_D_A_Z", "MFCC12_E_D_A_Z", "PLP_E_D_A", "PLP_0_D_A_Z", "PLP_E_D_A_Z"]
KEYS = ["u2_16000", "u0_8000", "u10_8000", "u7_400", "u7_720", "u7_1040"]


@pytest.fixture(scope="module")
def golden_variants():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "htk_variants_synth.npz"))


@pytest.mark.parametrize("name", NAMES)
def test_variant_bit_exact_with_reference_fft(oracle, golden_variants, name):
    if not oracle.use_reference_fft(True):
        pytest.skip("oracle/_ref/libref_dsp.so not built")
    try:
        for k in KEYS:
            out = oracle.htk_variant_chain(name, golden_variants["pcm_" + k])
            ref = golden_variants[name + "_" + k]
            assert out.shape == ref.shape, (name, k)
            assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), (name, k, np.abs(out - ref).max())
    finally:
        oracle.use_reference_fft(False)


def variant_tolerance(out, ref, n_cep, what="", raw=None):
    """Own / HIP FFT: 1e-5 of the frame's largest cepstral magnitude (the energy column: of its own value). For the
    mean-normalised (_Z) sets the magnitude is that of the frame BEFORE the mean was removed (raw: the sibling set's
    static block), since that is the quantity the FFT round-off scales with."""
    assert out.shape == ref.shape, what
    if not out.size:
        return
    D = ref.shape[1] // 3
    base = ref if raw is None else raw
    scale = np.maximum(np.abs(base[:, :n_cep]).max(axis=1, keepdims=True), 1.0)
    for blk in range(3):
        d = np.abs(out[:, blk * D:blk * D + n_cep].astype(np.float64) - ref[:, blk * D:blk * D + n_cep])
        assert (d / scale).max() <= 1e-5, f"{what}: block {blk}"
        if D > n_cep:
            e = np.abs(out[:, blk * D + n_cep].astype(np.float64) - ref[:, blk * D + n_cep])
            assert (e / np.maximum(np.abs(ref[:, n_cep]), 1.0)).max() <= 1e-6, f"{what}: energy, block {blk}"


@pytest.mark.parametrize("name", NAMES)
def test_variant_builtin_fft_bit_exact(oracle, golden_variants, name):
    oracle.use_reference_fft(False)
    _, plp, energy, _ = oracle.HTK_VARIANTS[name]
    n_cep = (5 if plp else 12) + (0 if energy else 1)
    for k in KEYS:
        pcm = golden_variants["pcm_" + k]
        raw = oracle.htk_variant_chain(name[:-2], pcm) if name.endswith("_Z") else None
        from tolerance import assert_bits_equal
        assert_bits_equal(oracle.htk_variant_chain(name, pcm), golden_variants[name + "_" + k], f"{name} {k}")
