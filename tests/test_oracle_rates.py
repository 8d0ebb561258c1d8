"""The CPU oracle at sample rates other than 16 kHz (oracle/lld_oracle.c: lldo_set_sample_rate), pinned level by level against the
REAL binary run on the same file at that rate (oracle/_ref/SMILExtract; every frame size of the shipped configs is given in
seconds, so the same conf applies). tests/test_gpu_rates.py then holds the device against this oracle."""
import numpy as np
import pytest

from oracle import lldo

pytestmark = pytest.mark.skipif(not lldo.have_ref(), reason="oracle/_ref not built")

RATES = [8000, 11025, 22050, 32000, 44100, 48000]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture()
def at_rate():
    yield lldo.set_sample_rate
    lldo.set_sample_rate(16000)


@pytest.mark.parametrize("fs", RATES)
def test_f0_group_levels_at_rate(fs, at_rate):
    """cSpecScale (hps), cPitchShs (shs), the 60 ms energy, the Viterbi-smoothed pitch and cPitchJitter's four values."""
    from opensmile_amd import synth
    pcm = synth.utterance(21 + fs % 5, int(1.2 * fs) + 3, fs)
    ref = lldo.run_reference_taps(pcm, names=("pitch", "shs", "hps", "e60", "jit"), fs=fs)
    at_rate(fs)
    lldo.use_reference_fft(False)
    out, taps = lldo.compare_f0_chain(pcm, taps=True)
    N, H = ref_frame(fs, 0.060), ref_frame(fs, 0.010)
    jit = lldo.pitch_jitter(pcm, ref["pitch"][:, 0], N=N, H=H, fs=float(fs))
    assert (ref["pitch"][:, 0] > 0).sum() > 20
    for k in ("hps", "shs", "e60"):
        assert taps[k].shape == ref[k].shape and np.array_equal(bits(taps[k]), bits(ref[k])), k
    assert out.shape == ref["pitch"].shape and np.array_equal(bits(out), bits(ref["pitch"]))
    assert jit.shape == ref["jit"].shape and np.array_equal(bits(jit), bits(ref["jit"]))


def ref_frame(fs, sec):
    """cFramer's frame size / step in samples (winToVecProcessor.cpp:435-456: round(sec / T), T = 1.0 / fs as a double)."""
    import math
    return int(math.floor(sec / (1.0 / fs) + 0.5))


@pytest.mark.parametrize("fs", RATES)
def test_compare16_lld_level_at_rate(fs, at_rate):
    from opensmile_amd import synth
    pcm = synth.utterance(3 + fs % 3, int(1.5 * fs) + 11, fs)
    ref = lldo.run_reference_lld("compare16/ComParE_2016.conf", pcm, fs=fs)
    at_rate(fs)
    lldo.use_reference_fft(False)
    out = lldo.compare_lld_chain(pcm)
    assert out.shape == ref.shape == (ref.shape[0], 130) and np.array_equal(bits(out), bits(ref))


@pytest.mark.parametrize("fs", [8000, 22050, 32000, 44100, 48000])     # (11 025 Hz: see tests/test_gpu_rates.py)
def test_egemaps_lld_and_functionals_at_rate(fs, at_rate):
    from opensmile_amd import synth
    pcm = synth.utterance(4 + fs % 3, int(1.5 * fs) + 11, fs)
    ref = lldo.run_reference_egemaps(pcm, fs=fs)
    at_rate(fs)
    lldo.use_reference_fft(False)
    lld, fn = lldo.egemaps_lld_chain(pcm), lldo.egemaps_func(pcm)
    assert lld.shape == ref["lld"].shape and np.array_equal(bits(lld), bits(ref["lld"]))
    assert fn.shape == ref["func"].shape == (1, 88) and np.array_equal(bits(fn), bits(ref["func"]))


@pytest.mark.parametrize("fs", RATES)
def test_is09_lld_level_at_rate(fs, at_rate):
    from opensmile_amd import synth
    pcm = synth.utterance(5 + fs % 3, int(1.2 * fs) + 3, fs)
    ref = lldo.run_reference_lld("is09-13/IS09_emotion.conf", pcm, fs=fs)
    at_rate(fs)
    lldo.use_reference_fft(False)
    out = lldo.is09_chain(pcm)
    assert out.shape == ref.shape and np.array_equal(bits(out), bits(ref))
