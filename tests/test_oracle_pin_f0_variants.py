"""The option variants of the F0 front end beyond ComParE_2016's (oracle/lld_oracle_f0.c: lldo_specscale_init_ex / _frame_ex,
lldo_pitch_shs with n_cand and old_peaks) pinned against the real binary on the instances of the shipped files that use them:
IS10_paraling_compat and emobase2010 (cSpecScale without enhancement / smoothing / auditory weighting; cPitchShs with three candidates,
the older candidate picker, voicingCutoff 0.75, no F0raw / voicingClip), IS11_speaker_state (four candidates, older picker),
IS10_paraling (minF = 20, six candidates, greedy picker). Input level in, output levels out, bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from oracle import lldo

pytestmark = pytest.mark.skipif(not lldo.have_ref(), reason="oracle/_ref not built")


def same(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))))


def taps(conf, levels, pcm, td):
    wav = os.path.join(td, "in.wav")
    lldo.write_wav(wav, pcm, 16000)
    c = os.path.join(td, "t.conf")
    txt = "\\{%s}\n[componentInstances:cComponentManager]\n" % os.path.join(lldo.REF_DIR, "config", conf)
    txt += "".join("instance[tap_%s].type=cHtkSink\n" % l for l in levels)
    txt += "".join("[tap_%s:cHtkSink]\nreader.dmLevel=%s\nfilename=%s/tap_%s.htk\n" % (l, l, td, l) for l in levels)
    open(c, "w").write(txt)
    subprocess.run([os.path.join(lldo.REF_DIR, "SMILExtract"), "-C", c, "-I", wav, "-O", os.path.join(td, "o.bin"), "-l", "0"], cwd=td,
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return [lldo.read_htk(os.path.join(td, "tap_%s.htk" % l))[0] for l in levels]


# conf -> (levels mag / hps / shs, minF, cSpecScale flags, nCandidates, old picker, voicingCutoff, F0raw + voicingClip columns)
CASES = {
    "is09-13/IS10_paraling_compat.conf": (("fftmag40", "hps", "pitchShs"), 25.0, 0, 3, 1, 0.75, False),
    "emobase/emobase2010.conf": (("fftmag40", "hps", "pitchShs"), 25.0, 0, 3, 1, 0.75, False),
    "is09-13/IS11_speaker_state.conf": (("fftmagG60", "hpsG60", "pitchShsG60"), 25.0, 7, 4, 1, 0.7, True),
    "is09-13/IS10_paraling.conf": (("is10_fftmag40", "is10_hps", "is10_pitchShs"), 20.0, 7, 6, 0, 0.7, False),
}


@pytest.mark.parametrize("conf", sorted(CASES))
def test_specscale_and_pitchshs_variants_bit_exact(conf, tmp_path):
    from opensmile_amd import synth
    levels, min_f, flags, nc, old, cutoff, tail = CASES[conf]
    for u, n in ((9, 32000), (71, 24000)):
        mag, hps_ref, shs_ref = taps(conf, levels, synth.utterance(u, n), str(tmp_path))
        K = mag.shape[1]
        hps, shs = lldo.specscale_shs_rows(mag, (K - 1) * 2 / 16000.0, min_f, flags, nc, old, cutoff)
        assert same(hps, hps_ref), (conf, u)
        assert same(shs if tail else shs[:, :1 + 3 * nc], shs_ref), (conf, u)
