#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference
binary (oracle/_ref/SMILExtract, built from /root/reference by oracle/Makefile)
on inputs from the synthetic-corpus contract (opensmile_amd/synth.py).

Run from the repo root in a container that has /root/reference:
    python tests/golden/make_golden.py
The .npz files are committed; the GPU box only reads them.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lldo  # noqa: E402
from opensmile_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def gen_func():
    # IS09_emotion functionals (func level, 384 = 32 x 12) together with the LLD level they summarise
    ref = {}
    for name, (u, n) in {"u2_32000": (2, 32000), "u3_16000": (3, 16000), "u10_16000": (10, 16000), "u1_16000": (1, 16000),
                          "u0_16000": (0, 16000), "u7_400": (7, 400), "u7_560": (7, 560), "u5_160000": (5, 160000)}.items():
        pcm = synth.utterance(u, n)
        f, x = lldo.run_reference_func("is09-13/IS09_emotion.conf", pcm)
        ref["pcm_" + name] = pcm
        ref["lld_" + name] = x
        ref["func_" + name] = f
        print("is09 func", name, x.shape, f.shape)
    np.savez_compressed(os.path.join(OUT, "is09_func_synth.npz"), **ref)




def gen_htk_variants():
    """The six sibling configs of config/mfcc and config/plp (log-energy column, cepstral mean subtraction)."""
    ref = {}
    for name, (conf, _, _, _) in lldo.HTK_VARIANTS.items():
        if name in ("MFCC12_0_D_A", "PLP_0_D_A"):
            continue
        for key, (u, n) in {"u2_16000": (2, 16000), "u0_8000": (0, 8000), "u10_8000": (10, 8000), "u7_400": (7, 400),
                            "u7_720": (7, 720), "u7_1040": (7, 1040)}.items():
            pcm = synth.utterance(u, n)
            ref["pcm_" + key] = pcm
            ref[name + "_" + key] = lldo.run_reference(conf, pcm)
        print("variant", name)
    np.savez_compressed(os.path.join(OUT, "htk_variants_synth.npz"), **ref)

def gen_f0():
    """ComParE_2016 F0 group: the levels of oracle/conf/compare_f0_taps.conf from the real binary
    (pitch = is13_pitchG60, shs = is13_pitchShsG60, vit = is13_pitchG60_viterbi, e60 = is13_e60,
    jit = is13_jitterShimmer, nzsmo / nzsmo_de = is13_lld_nzsmo[_de], lld = the 130-column LLD level;
    hps = is13_hpsG60 only for two short inputs, it is 513 columns wide)."""
    ref = {}
    for name, (u, n) in {"u2_16000": (2, 16000), "u3_16000": (3, 16000), "u10_16000": (10, 16000),
                          "u1_16000": (1, 16000), "u0_16000": (0, 16000), "u7_960": (7, 960), "u7_1120": (7, 1120),
                          "u7_1600": (7, 1600), "u4_48000": (4, 48000), "u11_160000": (11, 160000),
                          # ends with 5 / 4 / 3 frames the Viterbi smoother had not decided (end-of-input phases)
                          "u4_9000": (4, 9000), "u37_9000": (37, 9000), "u2_8720": (2, 8720)}.items():
        pcm = synth.utterance(u, n)
        t = lldo.run_reference_taps(pcm)
        ref["pcm_" + name] = pcm
        for k in ("pitch", "shs", "vit", "e60", "jit", "nzsmo", "nzsmo_de"):
            ref[k + "_" + name] = t[k]
        ref["lldf0_" + name] = np.concatenate([t["lld"][:, 0:6], t["lld"][:, 65:71]], axis=1)
        if name in ("u2_16000", "u4_9000", "u37_9000", "u2_8720", "u7_1600", "u10_16000"):
            ref["lld130_" + name] = t["lld"]          # the whole LLD level (groups A+B are also in compare16_ab_synth.npz)
        if name in ("u2_16000", "u7_1120"):
            ref["hps_" + name] = t["hps"]
        print("f0", name, t["pitch"].shape, t["lld"].shape)
    np.savez_compressed(os.path.join(OUT, "compare16_f0_synth.npz"), **ref)

def gen_func16():
    """ComParE_2016 functionals: the 6373-value vector of the real binary plus, from the taps of
    oracle/conf/compare_func_taps.conf, every row of the levels its six cFunctionals instances read."""
    ref = {}
    names = None
    for name, (u, n) in {"u3_48000": (3, 48000), "u5_16000": (5, 16000), "u4_9000": (4, 9000), "u37_9000": (37, 9000),
                          "u2_8720": (2, 8720), "u7_2720": (7, 2720), "u7_1760": (7, 1760), "u7_1440": (7, 1440)}.items():
        pcm = synth.utterance(u, n)
        t = lldo.run_reference_func_taps(pcm)
        ref["pcm_" + name] = pcm
        ref["func_" + name] = t["func"]
        for k in lldo.FUNC_TAPS:
            ref[k + "_" + name] = t[k]
        names = names or t["names"]
        assert names == t["names"]
        print("func16", name, t["func"].shape, t["b_smo"].shape)
    ref["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "compare16_func_synth.npz"), **ref)


def gen_is13():
    """config/is09-13/IS13_ComParE.conf on the real binary: the 130-column LLD level and the 6373 functionals (+ the rows
    of the levels its functionals read, for the two short inputs)."""
    ref = {}
    for name, (u, n) in {"u3_48000": (3, 48000), "u4_9000": (4, 9000), "u7_1760": (7, 1760), "u10_16000": (10, 16000)}.items():
        pcm = synth.utterance(u, n)
        t = lldo.run_reference_func_taps(pcm, is13=True)
        ref["pcm_" + name] = pcm
        ref["lld130_" + name] = t["lld"]
        ref["func_" + name] = t["func"]
        if n <= 9000:
            for k in lldo.FUNC_TAPS:
                ref[k + "_" + name] = t[k]
        print("is13", name, t["lld"].shape, t["func"].shape)
    np.savez_compressed(os.path.join(OUT, "is13_compare_synth.npz"), **ref)


def gen_egemaps_func():
    """eGeMAPSv02.conf on the real binary: the 88 functionals (summaries of LLDs this repo does not compute itself --
    used to check the plugin's cFunctionals override on the instances of the GeMAPS sets: Moments with stddevNorm,
    3-point percentiles, Peaks2 slopes / numPeaks, Segments nonX / eqX in seconds, nonZeroFuncts)."""
    ref = {}
    for name, (u, n) in {"u3_48000": (3, 48000), "u2_32000": (2, 32000), "u4_16000": (4, 16000), "u10_16000": (10, 16000)}.items():
        pcm = synth.utterance(u, n)
        f, _ = lldo.run_reference_func("egemaps/v02/eGeMAPSv02.conf", pcm)
        ref["pcm_" + name] = pcm
        ref["func_" + name] = f
        print("egemaps", name, f.shape)
    np.savez_compressed(os.path.join(OUT, "egemaps_func_synth.npz"), **ref)


def gen_egemaps():
    """BASELINE config 5, config/egemaps/v02/eGeMAPSv02.conf on the real binary: the 25-column LLD level, the 88
    functionals and HTK taps on the internal levels (oracle/conf/egemaps_taps.conf; keys of lldo.EGEMAPS_LEVELS).
    The cases cover T60 = 1 .. 995 and the end-of-input situations of the graph: P (frames the Viterbi smoother had not
    decided at the end) = 1 .. 9, P == T60, T60 - P = 2 .. 4, silence, noise, clipping square wave."""
    ref = {}
    lv = ("loudness", "lspec", "flux", "mfcc", "energy2", "formants", "lpc", "pitch", "jitter", "harm", "shs", "e60",
          "E", "F", "logf0", "loud", "NoZ", "NoNz", "specV", "specU")
    for name, (u, n) in {"u2_16000": (2, 16000), "u3_48000": (3, 48000), "u10_16000": (10, 16000), "u1_16000": (1, 16000),
                          "u0_16000": (0, 16000), "u7_960": (7, 960), "u7_1120": (7, 1120), "u7_1600": (7, 1600),
                          "u7_2720": (7, 2720), "u4_9000": (4, 9000), "u37_9000": (37, 9000), "u2_8720": (2, 8720),
                          "u5_16000": (5, 16000), "u11_160000": (11, 160000), "u3_1600": (3, 1600), "u3_1760": (3, 1760),
                          "u10_1280": (10, 1280), "u2_1760": (2, 1760), "u28_2240": (28, 2240), "u4_1920": (4, 1920),
                          "u10_1440": (10, 1440), "u3_1280": (3, 1280), "u7_800": (7, 800)}.items():
        pcm = synth.utterance(u, n)
        t = lldo.run_reference_egemaps(pcm, levels=lv)
        ref["pcm_" + name] = pcm
        ref["lld_" + name] = t["lld"]
        ref["func_" + name] = t["func"]
        for k in lv:
            ref[k + "_" + name] = t[k]
        print("egemaps", name, t["lld"].shape, t["func"].shape)
    np.savez_compressed(os.path.join(OUT, "egemaps_lld_synth.npz"), **ref)


def gen_plp():
    # config/plp/PLP_0_D_A.conf (PLP-CC + delta + accel, 18 columns): R8's IDFT / LP / cepstrum branch
    ref = {}
    for name, (u, n) in {"u2_16000": (2, 16000), "u3_16000": (3, 16000), "u10_16000": (10, 16000), "u1_16000": (1, 16000),
                          "u0_16000": (0, 16000), "u7_399": (7, 399), "u7_400": (7, 400), "u7_560": (7, 560),
                          "u7_1000": (7, 1000), "u5_160000": (5, 160000)}.items():
        pcm = synth.utterance(u, n)
        y = lldo.run_reference("plp/PLP_0_D_A.conf", pcm)
        ref["pcm_" + name] = pcm
        ref["out_" + name] = y if y.size else np.zeros((0, 18), np.float32)
        print("plp", name, y.shape)
    np.savez_compressed(os.path.join(OUT, "plp_0_d_a_synth.npz"), **ref)


def main(only=None):
    lldo.build()
    assert lldo.have_ref(), "oracle/_ref/SMILExtract missing (needs /root/reference)"
    if only == "func":
        gen_func()
        return
    if only == "plp":
        gen_plp()
        return
    if only == "is13":
        gen_is13()
        return
    if only == "egemaps":
        gen_egemaps_func()
        return
    if only == "egemaps_lld":
        gen_egemaps()
        return
    if only == "func16":
        gen_func16()
        return
    # config 2 shape, shortened: utterances 0 (zeros), 1 (square), 2, 3 (voiced), 10 (noise)
    # at 1.0 s, plus ragged lengths around the frame boundary (399/400/401/559/560/561 samples)
    cases = {}
    for u in (0, 1, 2, 3, 10):
        pcm = synth.utterance(u, 16000)
        cases[f"u{u}_16000"] = pcm
    for n in (399, 400, 401, 559, 560, 561, 1000):
        cases[f"u7_{n}"] = synth.utterance(7, n)
    ref = {}
    for k, pcm in cases.items():
        y = lldo.run_reference("mfcc/MFCC12_0_D_A.conf", pcm)
        ref["pcm_" + k] = pcm
        ref["out_" + k] = y
        print(k, y.shape)
    np.savez_compressed(os.path.join(OUT, "mfcc12_0_d_a_synth.npz"), **ref)

    # config 3 shape (IS09_emotion LLD level, 16 LLD + 16 delta, T+1 rows), shortened
    ref = {}
    for name, (u, n) in {"u2_16000": (2, 16000), "u3_16000": (3, 16000), "u10_16000": (10, 16000),
                          "u1_16000": (1, 16000), "u0_16000": (0, 16000), "u7_399": (7, 399), "u7_400": (7, 400),
                          "u7_560": (7, 560), "u7_720": (7, 720), "u7_880": (7, 880), "u7_1040": (7, 1040),
                          "u4_48000": (4, 48000)}.items():
        pcm = synth.utterance(u, n)
        y = lldo.run_reference_lld("is09-13/IS09_emotion.conf", pcm)
        ref["pcm_" + name] = pcm
        ref["out_" + name] = y
        print("is09", name, y.shape)
    np.savez_compressed(os.path.join(OUT, "is09_lld_synth.npz"), **ref)

    # config 4 shape (ComParE_2016 LLD level), groups A+B = columns 6..64 and 71..129 of the 130-column file
    ref = {}
    for name, (u, n) in {"u2_16000": (2, 16000), "u3_16000": (3, 16000), "u10_16000": (10, 16000),
                          "u1_16000": (1, 16000), "u0_16000": (0, 16000), "u7_1440": (7, 1440), "u7_1600": (7, 1600),
                          "u4_48000": (4, 48000)}.items():
        pcm = synth.utterance(u, n)
        y = lldo.run_reference_lld("compare16/ComParE_2016.conf", pcm)
        ref["pcm_" + name] = pcm
        ref["out_" + name] = np.concatenate([y[:, 6:65], y[:, 71:130]], axis=1)
        print("compare", name, y.shape)
    np.savez_compressed(os.path.join(OUT, "compare16_ab_synth.npz"), **ref)

    gen_func()
    gen_plp()
    gen_f0()
    gen_htk_variants()
    gen_func16()
    gen_egemaps_func()
    gen_egemaps()
    gen_is13()

    # config 1: the reference's example wav (44.1 kHz) -> known answer of SURVEY.md §8(c)
    import wave
    wav = os.path.join(lldo.REF_DIR, "opensmile.wav")
    with wave.open(wav, "rb") as w:
        fs = w.getframerate()
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").copy()
    y = lldo.run_reference("mfcc/MFCC12_0_D_A.conf", pcm, fs=fs)
    print("config1", fs, pcm.shape, y.shape, y[0, :3])
    np.savez_compressed(os.path.join(OUT, "mfcc12_0_d_a_config1.npz"), out=y, fs=fs,
                        n_samples=len(pcm))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
