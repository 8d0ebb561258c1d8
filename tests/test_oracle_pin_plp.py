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


def plp_conf_cut(oracle, stage, td, tap=True):
    """config/plp/PLP_0_D_A.conf with [plp:cPlp] cut after a stage (1: doLP = 0, 2: doLpToCeps = 0) and its level tapped"""
    import os
    base = os.path.join(oracle.REF_DIR, "config", "plp", "PLP_0_D_A.conf")
    txt = open(base).read().replace("\\{../shared/", "\\{" + os.path.join(oracle.REF_DIR, "config", "shared") + "/")
    txt = txt.replace("doLpToCeps = 1", "doLpToCeps = 0")
    if stage == 1:
        txt = txt.replace("doLP = 1", "doLP = 0")
    if tap:
        txt += ("\n[componentInstances:cComponentManager]\ninstance[tap_plp].type=cHtkSink\n[tap_plp:cHtkSink]\nreader.dmLevel=plp\n"
                "filename=%s/tap_plp.htk\n" % td)
    c = os.path.join(td, "plp_cut%d.conf" % stage)
    open(c, "w").write(txt)
    return c


@pytest.mark.parametrize("stage", [1, 2])
def test_plp_partial_modes_bit_exact(oracle, stage, tmp_path):
    """Round 6: cPlp's partial modes -- the autocorrelation (doIDFT = 1, doLP = 0) and the LP coefficients (doLP = 1, doLpToCeps = 0) as
    the component's output (plp.cpp:573-583) -- oracle/lld_oracle_compare.c::lldo_plp_stage against the REAL binary on PLP_0_D_A.conf
    with those two options changed, the [plp] level tapped: bit for bit."""
    import os
    import subprocess
    from opensmile_amd import synth
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built")
    td = str(tmp_path)
    c = plp_conf_cut(oracle, stage, td)
    for u, n in ((2, 16000), (7, 1000), (5, 48000)):
        pcm = synth.utterance(u, n)
        wav = os.path.join(td, "in.wav")
        oracle.write_wav(wav, pcm, 16000)
        subprocess.run([os.path.join(oracle.REF_DIR, "SMILExtract"), "-C", c, "-I", wav, "-O", os.path.join(td, "o.htk"), "-l", "0"], cwd=td,
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        ref = oracle.read_htk(os.path.join(td, "tap_plp.htk"))[0]
        got = oracle.plp_static_stage(pcm, stage)
        assert got.shape == ref.shape == (got.shape[0], 6 if stage == 1 else 5)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (stage, u, n)
