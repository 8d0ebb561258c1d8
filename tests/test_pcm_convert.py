"""R0 in every sample format (smilePcm_convertSamples, smileUtil.c:2500-2627): the oracle is
pinned bit-exact against the REAL function (oracle/_ref/libref_dsp.so is compiled from the
reference's smileUtil.c); the HIP kernel must equal the oracle bit for bit."""
import numpy as np
import pytest

FORMATS = [(1, 8), (2, 16), (3, 24), (4, 24), (4, 32)]


def raw_bytes(bps, ch, n, seed):
    rng = np.random.default_rng(seed)
    b = rng.integers(0, 256, size=bps * ch * n, dtype=np.uint8)
    # make the extremes appear: all-zero, all-ones, sign bit patterns
    b[: bps * ch] = 0
    b[bps * ch: 2 * bps * ch] = 255
    b[2 * bps * ch: 3 * bps * ch] = 128
    return b.tobytes()


@pytest.mark.parametrize("bps,bits", FORMATS)
@pytest.mark.parametrize("ch", [1, 2, 5])
@pytest.mark.parametrize("mix", [True, False])
def test_oracle_equals_real_function(oracle, bps, bits, ch, mix):
    raw = raw_bytes(bps, ch, 4096, 11 + bps + ch)
    ref = oracle.ref_pcm_convert(raw, bps, bits, ch, mix)
    if ref is None:
        pytest.skip("oracle/_ref/libref_dsp.so not built")
    out = oracle.pcm_convert(raw, bps, bits, ch, mix)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))


def test_known_answers(oracle):
    pcm = np.array([32767, -32767, -32768, 0, 1], dtype="<i2")
    x = oracle.pcm_convert(pcm.tobytes(), 2, 16, 1)
    assert x[0] == 1.0 and x[1] == -1.0 and x[2] < -1.0 and x[3] == 0.0      # 32767, not 32768
    st = np.array([[100, 300], [-5, 5]], dtype="<i2")
    assert np.allclose(oracle.pcm_convert(st.tobytes(), 2, 16, 2), [200 / 32767, 0.0])
    assert oracle.pcm_convert(st.tobytes(), 2, 16, 2, mixdown=False).shape == (2, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("bps,bits", FORMATS)
def test_hip_kernel_bit_exact(oracle, bps, bits):
    from opensmile_amd import capi
    ctx = capi.Context(0)
    for ch in (1, 2, 3):
        for mix in (True, False):
            raw = raw_bytes(bps, ch, 10007, 5 + bps + ch)
            out = capi.pcm_convert_host(ctx, raw, bps, bits, ch, mix)
            ref = oracle.pcm_convert(raw, bps, bits, ch, mix)
            assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), (bps, bits, ch, mix)


@pytest.mark.gpu
def test_hip_rejects_unknown_format():
    from opensmile_amd import capi
    ctx = capi.Context(0)
    with pytest.raises(capi.SmileHipError):
        capi.pcm_convert_host(ctx, b"\\0" * 40, 5, 40, 1)
    with pytest.raises(capi.SmileHipError):
        capi.pcm_convert_host(ctx, b"\\0" * 40, 4, 16, 1)
