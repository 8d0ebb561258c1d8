"""cMfcc with inverse = 1 (mfcc.cpp:184-235; round 6) -- oracle/lld_oracle_compare.c::lldo_mfcc_inverse against the REAL binary: a second
cMfcc instance behind MFCC12_0_D_A.conf's (or MFCC12_E_D_A.conf's: firstMfcc = 1) cepstral level turns it back into 26 mel bands; both
levels tapped, bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from oracle import lldo

pytestmark = pytest.mark.skipif(not lldo.have_ref(), reason="oracle/_ref not built")

# (file, level the cepstra are on, firstMfcc, lastMfcc, cepLifter, htkcompatible, doLog of the inverse instance)
CASES = {"htk_c0": ("MFCC12_0_D_A.conf", "ft0", 0, 12, 22, 1, 1), "htk_first1": ("MFCC12_E_D_A.conf", "mfcc", 1, 12, 22, 1, 1),
         "htk_nolog": ("MFCC12_0_D_A.conf", "ft0", 0, 12, 22, 1, 0)}


def inverse_conf(case, td):
    f, level, first, last, lifter, htk, dolog = CASES[case]
    base = os.path.join(lldo.REF_DIR, "config", "mfcc", f)
    txt = open(base).read().replace("\\{../shared/", "\\{" + os.path.join(lldo.REF_DIR, "config", "shared") + "/")
    txt += ("\n[componentInstances:cComponentManager]\ninstance[imel].type=cMfcc\ninstance[tap_c].type=cHtkSink\ninstance[tap_m].type=cHtkSink\n"
            "[imel:cMfcc]\nreader.dmLevel=%s\nwriter.dmLevel=imel\ninverse=1\nnBands=26\nfirstMfcc=%d\nlastMfcc=%d\ncepLifter=%d\nhtkcompatible=%d\n"
            "doLog=%d\n[tap_c:cHtkSink]\nreader.dmLevel=%s\nfilename=%s/tap_c.htk\n[tap_m:cHtkSink]\nreader.dmLevel=imel\nfilename=%s/tap_m.htk\n"
            % (level, first, last, lifter, htk, dolog, level, td, td))
    c = os.path.join(td, case + ".conf")
    open(c, "w").write(txt)
    return c


@pytest.mark.parametrize("case", sorted(CASES))
def test_mfcc_inverse_bit_exact(case, tmp_path):
    from opensmile_amd import synth
    td = str(tmp_path)
    c = inverse_conf(case, td)
    _f, _level, first, last, lifter, htk, dolog = CASES[case]
    for u, n in ((2, 16000), (5, 48000)):
        wav = os.path.join(td, "in.wav")
        lldo.write_wav(wav, synth.utterance(u, n), 16000)
        subprocess.run([os.path.join(lldo.REF_DIR, "SMILExtract"), "-C", c, "-I", wav, "-O", os.path.join(td, "o.htk"), "-l", "0"], cwd=td,
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cep = lldo.read_htk(os.path.join(td, "tap_c.htk"))[0]
        ref = lldo.read_htk(os.path.join(td, "tap_m.htk"))[0]
        got = lldo.mfcc_inverse_rows(cep, first, last, 26, lifter, htk, dolog)
        assert got.shape == ref.shape == (cep.shape[0], 26)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (case, u, n)
