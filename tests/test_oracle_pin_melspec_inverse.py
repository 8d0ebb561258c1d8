"""cMelspec with inverse = 1 (melspec.cpp:466-516; round 6) -- oracle/lld_oracle_compare.c::lldo_melspec_inverse against the REAL binary: a
second cMelspec instance behind MFCC12_0_D_A.conf's mel level turns the 26 bands back into a 257-bin (or 129-bin) magnitude spectrum;
both levels tapped, bit for bit. Power and magnitude banks, HTK scaling on and off, a band edge inside the spectrum."""
import os
import subprocess

import numpy as np
import pytest

from oracle import lldo

pytestmark = pytest.mark.skipif(not lldo.have_ref(), reason="oracle/_ref not built")

# (bins to create, usePower, htkcompatible, lofreq, hifreq) of the inverse instance
CASES = {"power_htk_257": (257, 1, 1, 0.0, 8000.0), "mag_257": (257, 0, 1, 0.0, 8000.0), "power_nohtk_129": (129, 1, 0, 0.0, 8000.0),
         "power_htk_band": (257, 1, 1, 300.0, 6000.0)}


def inverse_conf(case, td):
    n_out, power, htk, lo, hi = CASES[case]
    base = os.path.join(lldo.REF_DIR, "config", "mfcc", "MFCC12_0_D_A.conf")
    txt = open(base).read().replace("\\{../shared/", "\\{" + os.path.join(lldo.REF_DIR, "config", "shared") + "/")
    txt += ("\n[componentInstances:cComponentManager]\ninstance[ispec].type=cMelspec\ninstance[tap_m].type=cHtkSink\ninstance[tap_s].type=cHtkSink\n"
            "[ispec:cMelspec]\nreader.dmLevel=melspec\nwriter.dmLevel=ispec\ninverse=1\nnBands=%d\nusePower=%d\nhtkcompatible=%d\nlofreq=%g\nhifreq=%g\n"
            "specScale=mel\n[tap_m:cHtkSink]\nreader.dmLevel=melspec\nfilename=%s/tap_m.htk\n[tap_s:cHtkSink]\nreader.dmLevel=ispec\nfilename=%s/tap_s.htk\n"
            % (n_out, power, htk, lo, hi, td, td))
    c = os.path.join(td, case + ".conf")
    open(c, "w").write(txt)
    return c


def run_ref(case, td, u, n):
    from opensmile_amd import synth
    c = inverse_conf(case, td)
    wav = os.path.join(td, "in.wav")
    lldo.write_wav(wav, synth.utterance(u, n), 16000)
    subprocess.run([os.path.join(lldo.REF_DIR, "SMILExtract"), "-C", c, "-I", wav, "-O", os.path.join(td, "o.htk"), "-l", "0"], cwd=td,
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return lldo.read_htk(os.path.join(td, "tap_m.htk"))[0], lldo.read_htk(os.path.join(td, "tap_s.htk"))[0]


@pytest.mark.parametrize("case", sorted(CASES))
def test_melspec_inverse_bit_exact(case, tmp_path):
    n_out, power, htk, lo, hi = CASES[case]
    for u, n in ((2, 16000), (5, 48000)):
        mel, ref = run_ref(case, str(tmp_path), u, n)
        got = lldo.melspec_inverse_rows(mel, n_out, 512 / 16000.0, lo, hi, power, htk)   # (the level keeps the frame size cTransformFFT gave it: 512 samples)
        assert got.shape == ref.shape == (mel.shape[0], n_out)
        assert np.abs(ref).max() > 0
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (case, u, n, int((got.view(np.uint32) != ref.view(np.uint32)).sum()))
