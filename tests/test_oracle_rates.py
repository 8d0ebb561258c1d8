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
