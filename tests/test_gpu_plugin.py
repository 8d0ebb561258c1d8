"""Drop-in boundary: the UNMODIFIED reference binary (oracle/_ref/SMILExtract,
shared build) runs the UNMODIFIED config/mfcc/MFCC12_0_D_A.conf with
opensmile_amd/plugin/plugins/libsmilehip_plugin.so in ./plugins: the six chain
components are replaced by name and push every frame through the HIP kernels."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from tolerance import assert_parity, corpus_col_scale

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLUGDIR = os.path.join(ROOT, "opensmile_amd", "plugin")


def _run(oracle, pcm, env_extra=None, conf="mfcc/MFCC12_0_D_A.conf", out_opt="-O", fs=16000):
    exe = os.path.join(oracle.REF_DIR, "SMILExtract")
    plug = os.path.join(PLUGDIR, "plugins", "libsmilehip_plugin.so")
    if not (os.path.exists(exe) and os.path.exists(plug)):
        pytest.skip("oracle/_ref/SMILExtract or the plugin .so not built (needs /root/reference at build time)")
    with tempfile.TemporaryDirectory() as td:
        wav, out, trace = (os.path.join(td, n) for n in ("in.wav", "out.htk", "trace.txt"))
        oracle.write_wav(wav, pcm, fs)
        env = dict(os.environ)
        env["LD_LIBRARY_PATH"] = os.pathsep.join(
            [os.path.join(ROOT, "opensmile_amd"), oracle.REF_DIR, env.get("LD_LIBRARY_PATH", "")])
        env["SMILEHIP_PLUGIN_TRACE"] = trace
        env.update(env_extra or {})
        # cwd = the directory that contains ./plugins (componentManager.cpp:347-364)
        r = subprocess.run([exe, "-C", conf if os.path.isabs(conf) else os.path.join(oracle.REF_DIR, "config", conf), "-I", wav, out_opt, out,
                            "-l", "1"], cwd=PLUGDIR, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert os.path.exists(out), r.stderr[-2000:]
        y = oracle.read_htk(out)[0]
        tr = dict(l.split() for l in open(trace).read().split("\n") if l.strip()) if os.path.exists(trace) else {}
    return y, {k: int(v) for k, v in tr.items()}


def test_plugin_runs_unmodified_config(oracle, golden_synth):
    keys = ["u2_16000", "u10_16000", "u0_16000"]
    s_col = corpus_col_scale([golden_synth["out_" + k] for k in keys], 13)
    for k in keys:
        y, tr = _run(oracle, golden_synth["pcm_" + k])
        ref = golden_synth["out_" + k]
        # every overridden component saw every frame (98 frames; the FFT etc. run once per frame)
        for comp in ("cVectorPreemphasis", "cWindower", "cTransformFFT", "cFFTmagphase", "cMelspec", "cMfcc"):
            assert tr.get(comp, 0) == ref.shape[0], f"{comp} not routed through the plugin: {tr}"
        assert_parity(y, ref, block=13, what=f"plugin {k}", col_scale=s_col)


def test_plugin_exact_stages_are_bit_exact(oracle, golden_synth):
    """Overriding only the stages whose arithmetic is the reference's own float
    sequence (R2, R3, R5, R6) must reproduce the reference BIT FOR BIT; adding
    R7 (libm logf vs the kernel's correctly rounded log: <= 1 ulp on a few
    log-mel values) stays within 1e-6 of the frame's largest coefficient."""
    for k in ("u3_16000", "u10_16000"):
        ref = golden_synth["out_" + k]
        y, tr = _run(oracle, golden_synth["pcm_" + k],
                     {"SMILEHIP_PLUGIN_COMPONENTS": "cVectorPreemphasis,cWindower,cFFTmagphase,cMelspec"})
        assert tr["cTransformFFT"] == 0 and tr["cMfcc"] == 0 and tr["cMelspec"] == ref.shape[0]
        assert np.array_equal(y.view(np.uint32), ref.view(np.uint32)), f"max abs {np.abs(y - ref).max()}"
        y, tr = _run(oracle, golden_synth["pcm_" + k],
                     {"SMILEHIP_PLUGIN_COMPONENTS": "cVectorPreemphasis,cWindower,cFFTmagphase,cMelspec,cMfcc"})
        assert tr["cMfcc"] == ref.shape[0]
        scale = np.abs(ref[:, :13]).max(axis=1, keepdims=True)
        assert (np.abs(y - ref) / np.tile(np.maximum(scale, 1e-30), (1, 1))).max() < 1e-6


IS09 = "is09-13/IS09_emotion.conf"
IS09_COMPS = ("cVectorPreemphasis", "cWindower", "cTransformFFT", "cFFTmagphase", "cMelspec", "cMfcc", "cEnergy", "cMZcr",
              "cAcf", "cPitchACF", "cDeltaRegression", "cContourSmoother")


def test_plugin_is09_exact_components_are_bit_exact(oracle, golden_is09):
    """IS09_emotion.conf, unmodified, with the components whose arithmetic is the reference's own
    sequence replaced: cMZcr (integer count), cContourSmoother and cDeltaRegression (float stencils
    through cWindowProcessor::processBuffer, incl. its end-of-input padding) and cPitchACF's scalar
    smoother around the device's peak pick must reproduce the binary's LLD file bit for bit;
    cEnergy (parallel double sum) to float round-off."""
    for k in ("u2_16000", "u7_560", "u7_400"):
        ref = golden_is09["out_" + k]
        y, tr = _run(oracle, golden_is09["pcm_" + k],
                     {"SMILEHIP_PLUGIN_COMPONENTS": "cMZcr,cContourSmoother,cDeltaRegression,cPitchACF"}, IS09, "-lldhtkoutput")
        assert y.shape == ref.shape
        T = ref.shape[0] - 1
        assert tr["cMZcr"] == T and tr["cPitchACF"] == T and tr["cWindower"] == 0
        assert tr["cContourSmoother"] == 16 * (T + 1) and tr["cDeltaRegression"] == 16 * (T + 3)   # rows x elements incl. EOI rows
        assert np.array_equal(y.view(np.uint32), ref.view(np.uint32)), f"{k}: max abs {np.abs(y - ref).max()}"
    ref = golden_is09["out_u3_16000"]
    y, tr = _run(oracle, golden_is09["pcm_u3_16000"], {"SMILEHIP_PLUGIN_COMPONENTS": "cEnergy"}, IS09, "-lldhtkoutput")
    assert tr["cEnergy"] == ref.shape[0] - 1
    assert np.abs(y[:, 0] - ref[:, 0]).max() <= 1e-7 * np.abs(ref[:, 0]).max()
    assert np.array_equal(y[:, 1:16], ref[:, 1:16])


def test_plugin_is09_all_twelve(oracle, golden_is09):
    """Every §8(a) component of the IS09 chain behind the reference's own operator API."""
    from test_gpu_is09 import check_lld
    for k in ("u2_16000", "u10_16000", "u0_16000"):
        ref = golden_is09["out_" + k]
        y, tr = _run(oracle, golden_is09["pcm_" + k], None, IS09, "-lldhtkoutput")
        T = ref.shape[0] - 1
        for comp in IS09_COMPS:
            assert tr.get(comp, 0) >= T, f"{comp} not routed through the plugin: {tr}"
        assert tr["cAcf"] == 2 * T                       # the ACF and the cepstrum instance
        check_lld(y, ref, f"plugin is09 {k}")


COMPARE = "compare16/ComParE_2016.conf"


def _ab(y):
    return np.concatenate([y[:, 6:65], y[:, 71:130]], axis=1)


def test_plugin_compare_spectral_and_plp(oracle, golden_compare):
    """ComParE_2016.conf, unmodified: cSpectral and the two cPlp instances (auditory spectrum with and
    without newRASTA) behind the reference's operator API. Their inputs are the binary's own magnitudes
    and mel bands, so the outputs agree with the binary's LLD file to the round-off of the double-
    accumulated sums / pow / log / exp (no FFT difference involved)."""
    from test_oracle_pin_compare import compare_tolerances
    for k in ("u2_16000", "u10_16000"):
        ref = golden_compare["out_" + k]
        y, tr = _run(oracle, golden_compare["pcm_" + k], {"SMILEHIP_PLUGIN_COMPONENTS": "cSpectral,cPlp"}, COMPARE, "-lldhtkoutput")
        y = _ab(y)
        assert y.shape == ref.shape
        T20 = ref.shape[0] + 3                          # rows = T60 + 1, T20 = T60 + 4
        assert tr["cSpectral"] == T20 and tr["cPlp"] == 2 * T20 and tr["cMelspec"] == 0
        d = np.abs(y.astype(np.float64) - ref)
        scale = np.maximum(np.abs(ref[:, :59]).max(axis=0), 1e-12)
        rel = np.concatenate([d[:, :59] / scale, d[:, 59:] / scale], axis=1)
        ro = [32, 33, 34, 35, 59 + 32, 59 + 33, 59 + 34, 59 + 35]       # roll-off: threshold test on a prefix sum
        other = [c for c in range(118) if c not in ro]
        assert rel[:, other].max() <= 2e-6, f"{k}: col {other[int(rel[:, other].max(axis=0).argmax())]} rel {rel[:, other].max():.2e}"
        assert (d[:, ro] > 1e-3).mean() <= 0.01
        compare_tolerances(y, ref, k)


def test_plugin_compare_f0_components(oracle, golden_f0):
    """ComParE_2016.conf, unmodified: cSpecScale and cPitchShs behind the reference's operator API. Their input is the
    binary's own magnitude spectrum and everything after it (Viterbi smoother, jitter, smoothing, deltas) is the
    binary's code, so the F0 group's 12 LLD columns must come out (almost) bit-identical: the one libm call on the
    device path is exp() of the refined candidate frequency."""
    for k in ("u2_16000", "u4_9000"):
        ref = golden_f0["lld130_" + k]
        y, tr = _run(oracle, golden_f0["pcm_" + k], {"SMILEHIP_PLUGIN_COMPONENTS": "cSpecScale,cPitchShs"}, COMPARE, "-lldhtkoutput")
        assert y.shape == ref.shape
        T60 = ref.shape[0] - 1
        assert tr["cSpecScale"] == T60 and tr["cPitchShs"] == T60 and tr["cMelspec"] == 0
        cols = list(range(0, 6)) + list(range(65, 71))
        same = (y[:, cols].view(np.uint32) == ref[:, cols].view(np.uint32)).all(axis=1)
        assert same.mean() >= 0.97, f"{k}: {int((~same).sum())} of {len(same)} rows differ"
        scale = np.maximum(np.abs(ref[:, :6]).max(axis=0), 1e-6)
        assert (np.abs(y[:, cols] - ref[:, cols]) / np.concatenate([scale, scale])).max() <= 1e-4, k
        # everything else is untouched reference code
        rest = [c for c in range(130) if c not in cols]
        assert np.array_equal(y[:, rest].view(np.uint32), ref[:, rest].view(np.uint32))


def test_plugin_compare_all_overrides(oracle, golden_compare):
    from test_oracle_pin_compare import compare_tolerances
    ref = golden_compare["out_u3_16000"]
    y, tr = _run(oracle, golden_compare["pcm_u3_16000"], None, COMPARE, "-lldhtkoutput")
    for comp in ("cWindower", "cTransformFFT", "cFFTmagphase", "cMelspec", "cMfcc", "cEnergy", "cMZcr", "cSpectral", "cPlp",
                 "cDeltaRegression", "cContourSmoother", "cSpecScale", "cPitchShs"):
        assert tr.get(comp, 0) > 0, f"{comp} not routed through the plugin: {tr}"
    compare_tolerances(_ab(y), ref, "plugin compare all")


def test_plugin_is09_functionals(oracle, golden_func):
    """cFunctionals::doProcess behind the plugin: IS09_emotion's 384 functionals computed from the
    binary's own LLD contours (only cFunctionals overridden) equal the binary's func level (every contour is walked
    in the reference's order: bit for bit but for the last-bit cases the bound below allows)."""
    for k in ("u3_16000", "u7_560", "u2_32000"):
        ref = golden_func["func_" + k]
        y, tr = _run(oracle, golden_func["pcm_" + k], {"SMILEHIP_PLUGIN_COMPONENTS": "cFunctionals"}, IS09, "-htkoutput")
        assert y.shape == ref.shape == (1, 384)
        assert tr["cFunctionals"] == 32 and tr["cMfcc"] == 0          # one doProcess per LLD contour
        exact = np.array([n in (0, 1, 2, 3, 4) for n in range(12)] * 32)
        assert np.array_equal(y[0, exact], ref[0, exact]), k
        d = np.abs(y.astype(np.float64) - ref)
        assert (d <= 1e-5 * np.abs(ref) + 1e-12).all(), (k, float((d / np.maximum(np.abs(ref), 1e-30)).max()))
        assert (y.view(np.uint32) == ref.view(np.uint32)).mean() >= 0.97


def test_plugin_compare16_functionals_all_families(oracle):
    """ComParE_2016.conf, unmodified, with only cFunctionals behind the plugin: its six instances (Extremes, Means,
    Moments, Regression incl. the quadratic part, Percentiles, Times, Segments, Lpc, Peaks2; nonZeroFuncts; three time
    norms) are translated into specs and computed on the device from the binary's own LLD contours -- the 6373 values
    equal the plain binary's bit for bit, except the few that pass through libm (log / exp; 1e-6)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "compare16_func_synth.npz"))
    names = [str(n) for n in g["names"]]
    libm = ("_flatness", "_peakMeanRel", "_centroid", "_linregc1", "_qregc1", "_qregc2")
    soft = np.array([n.endswith(libm) for n in names])
    for k in ("u7_2720", "u5_16000", "u4_9000"):
        ref = g["func_" + k][None, :]
        y, tr = _run(oracle, g["pcm_" + k], {"SMILEHIP_PLUGIN_COMPONENTS": "cFunctionals"}, COMPARE, "-htkoutput")
        assert y.shape == ref.shape == (1, 6373)
        assert tr["cFunctionals"] == 8 + 110 + 12 + 1 + 59 + 59, tr       # one doProcess per input element
        same = y.view(np.uint32) == ref.view(np.uint32)
        assert same[0, ~soft].all(), (k, [names[i] for i in np.flatnonzero(~same[0] & ~soft)[:6]])
        err = np.abs(y.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-6)
        assert err.max() <= 1e-6, (k, names[int(err.argmax())], float(err.max()))


def test_plugin_is13_compare_functionals(oracle):
    """config/is09-13/IS13_ComParE.conf with only cFunctionals behind the plugin: the option translation picks up the
    IS13 values (no ratio limiting, signed centroid, normInputs = 0, normRegCoeff = 0) and reproduces the binary."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "is13_compare_synth.npz"))
    names = [str(n) for n in np.load(os.path.join(ROOT, "tests", "golden", "compare16_func_synth.npz"))["names"]]
    soft = np.array([n.endswith(("_flatness",)) for n in names])
    for k in ("u4_9000", "u10_16000"):
        ref = g["func_" + k][None, :]
        y, tr = _run(oracle, g["pcm_" + k], {"SMILEHIP_PLUGIN_COMPONENTS": "cFunctionals"}, "is09-13/IS13_ComParE.conf", "-htkoutput")
        assert y.shape == ref.shape == (1, 6373) and tr["cFunctionals"] == 249
        same = y.view(np.uint32) == ref.view(np.uint32)
        assert same[0, ~soft].all(), (k, [names[i] for i in np.flatnonzero(~same[0] & ~soft)[:6]])
        err = np.abs(y.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-6)
        assert err.max() <= 1e-6


def test_plugin_egemaps_functionals(oracle):
    """eGeMAPSv02.conf, unmodified, with only cFunctionals behind the plugin: the functionals instances of the GeMAPS sets
    (Moments with stddevNorm = 2, 20/50/80 percentiles + range, Peaks2 slopes and numPeaks per second, Segments nonX and
    eqX in seconds, nonZeroFuncts = 1, masterTimeNorm) run on the device from the binary's own LLD contours and give the
    plain binary's 88 values (libm-dependent ones 1e-6)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "egemaps_func_synth.npz"))
    for k in ("u3_48000", "u4_16000", "u10_16000"):
        ref = g["func_" + k]
        y, tr = _run(oracle, g["pcm_" + k], {"SMILEHIP_PLUGIN_COMPONENTS": "cFunctionals"}, "egemaps/v02/eGeMAPSv02.conf", "-htkoutput")
        assert y.shape == ref.shape == (1, 88)
        assert tr["cFunctionals"] >= 25, tr                              # every instance went through the override
        err = np.abs(y.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-6)
        assert err.max() <= 1e-6, (k, int(err.argmax()), float(err.max()), float(y[0, err.argmax()]), float(ref[0, err.argmax()]))
        assert (y.view(np.uint32) == ref.view(np.uint32)).mean() >= 0.95


def test_plugin_plp_cepstra_bit_exact(oracle, golden_plp):
    """config/plp/PLP_0_D_A.conf, unmodified, with cPlp (PLP cepstra: IDFT, Durbin, LP -> cepstra, lifter) and the
    delta components behind the plugin: same float sequence as the reference, pow through a correctly
    rounded double pow -> the binary's output is reproduced bit for bit (or to 1 ulp of libm's pow)."""
    for k in ("u2_16000", "u7_560"):
        ref = golden_plp["out_" + k]
        y, tr = _run(oracle, golden_plp["pcm_" + k], {"SMILEHIP_PLUGIN_COMPONENTS": "cPlp,cDeltaRegression"}, "plp/PLP_0_D_A.conf")
        assert y.shape == ref.shape and tr["cPlp"] == ref.shape[0] and tr["cMelspec"] == 0
        scale = np.abs(ref[:, :6]).max(axis=1, keepdims=True)
        assert (np.abs(y - ref) / scale).max() <= 1e-6
        assert (y.view(np.uint32) == ref.view(np.uint32)).mean() >= 0.95


def test_plugin_fused_source_component(oracle, golden_synth):
    """Fused mode behind the component API: cHipLldSource (a new component type of the plugin) replaces the wave
    source and the nine chain components of MFCC12_0_D_A.conf; the reference's own output section writes the file.
    Same rows, same HTK header and CSV layout as the unmodified config gives with the CPU chain."""
    exe = os.path.join(oracle.REF_DIR, "SMILExtract")
    plug = os.path.join(PLUGDIR, "plugins", "libsmilehip_plugin.so")
    if not (os.path.exists(exe) and os.path.exists(plug)):
        pytest.skip("oracle/_ref/SMILExtract or the plugin .so not built")
    from test_host_io import parse_csv
    for key, fset, conf_ref in (("u2_16000", "mfcc12_0_d_a", "mfcc/MFCC12_0_D_A.conf"), ("u3_16000", "plp_0_d_a", "plp/PLP_0_D_A.conf"),
                                ("u3_16000", "mfcc12_e_d_a", "mfcc/MFCC12_E_D_A.conf")):
        pcm = golden_synth["pcm_" + key]
        with tempfile.TemporaryDirectory() as td:
            wav = os.path.join(td, "in.wav")
            oracle.write_wav(wav, pcm)
            env = dict(os.environ)
            env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "opensmile_amd"), oracle.REF_DIR, env.get("LD_LIBRARY_PATH", "")])
            outs = {}
            for tag, conf, extra in (("hip", os.path.join(PLUGDIR, "conf", "MFCC12_0_D_A_hip.conf"), ["-featureSet", fset]),
                                     ("ref", os.path.join(oracle.REF_DIR, "config", conf_ref), [])):
                htk, csv = os.path.join(td, tag + ".htk"), os.path.join(td, tag + ".csv")
                e = dict(env)
                if tag == "ref":
                    e["SMILEHIP_PLUGIN_COMPONENTS"] = "none"          # the plain CPU chain
                r = subprocess.run([exe, "-C", conf, "-I", wav, "-O", htk, "-csvoutput", csv, "-instname", "utt", "-l", "1"] + extra,
                                   cwd=PLUGDIR, env=e, capture_output=True, text=True, timeout=300)
                assert r.returncode == 0 and os.path.exists(htk), r.stderr[-2000:]
                outs[tag] = (open(htk, "rb").read()[:12], oracle.read_htk(htk)[0], parse_csv(csv))
        (hh, xh, ch), (hr, xr, cr) = outs["hip"], outs["ref"]
        assert hh == hr and xh.shape == xr.shape                        # HTK header: rows, period, vector size, parmKind
        n_static = xr.shape[1] // 3
        scale = np.abs(xr[:, :n_static]).max(axis=1, keepdims=True)
        assert (np.abs(xh - xr) / scale).max() <= 1e-5
        assert ch[0] == cr[0] and ch[1] == cr[1] and ch[2].shape == cr[2].shape    # CSV head line, element names
        assert np.array_equal(ch[2][:, 0], cr[2][:, 0])                 # frameTime column


def test_plugin_fused_source_compare16_functionals(oracle):
    """Fused mode for a set that ends in functionals: cHipLldSource with featureSet = compare16_func replaces the wave
    source, all LLD components and the six cFunctionals instances of ComParE_2016.conf and writes the 6373-value vector
    into the level `func`; the reference's sinks write the files. ARFF attribute block, CSV head line and HTK header
    are the unmodified config's, byte for byte; the values follow the statistical bar of test_gpu_func16.py."""
    exe = os.path.join(oracle.REF_DIR, "SMILExtract")
    plug = os.path.join(PLUGDIR, "plugins", "libsmilehip_plugin.so")
    if not (os.path.exists(exe) and os.path.exists(plug)):
        pytest.skip("oracle/_ref/SMILExtract or the plugin .so not built")
    G = os.path.join(ROOT, "tests", "golden", "files")
    with tempfile.TemporaryDirectory() as td:
        env = dict(os.environ)
        env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "opensmile_amd"), oracle.REF_DIR, env.get("LD_LIBRARY_PATH", "")])
        arff, htk, csv = (os.path.join(td, n) for n in ("f.arff", "f.htk", "f.csv"))
        r = subprocess.run([exe, "-C", os.path.join(PLUGDIR, "conf", "ComParE_2016_func_hip.conf"), "-I", os.path.join(G, "u3_4000.wav"),
                            "-O", arff, "-htkoutput", htk, "-csvoutput", csv, "-instname", "u3", "-l", "1"],
                           cwd=PLUGDIR, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and os.path.exists(arff) and os.path.exists(htk), r.stderr[-2000:]
        got, ref = open(arff).read(), open(os.path.join(G, "compare16_func_u3.arff")).read()
        assert got.split("@data")[0] == ref.split("@data")[0]
        dg, dr = got.split("@data")[1].strip().split(","), ref.split("@data")[1].strip().split(",")
        assert len(dg) == len(dr) == 6375 and dg[0] == dr[0] == "u3" and dg[-1] == dr[-1]
        assert open(htk, "rb").read()[:12] == open(os.path.join(G, "compare16_func_u3.htk"), "rb").read()[:12]
        x, xr = oracle.read_htk(htk)[0], oracle.read_htk(os.path.join(G, "compare16_func_u3.htk"))[0]
        err = np.abs(x[0].astype(np.float64) - xr[0]) / np.maximum(np.abs(xr[0]), 1e-2)
        from tolerance import record
        record("plugin_func16_source", within_1em3=(err <= 1e-3).mean(), median=np.median(err))
        assert (err <= 1e-3).mean() >= 0.99 and np.median(err) <= 1e-6          # measured 0.9969 / 0
        head = open(csv).readline().strip().split(";")
        assert head[:2] == ["name", "frameTime"] and len(head) == 6375
        row = open(csv).read().split("\n")[1].split(";")
        assert row[0] == "'u3'" and float(row[1]) == 0.0
        # the IS13 variant of the set through the same component
        htk13 = os.path.join(td, "f13.htk")
        r = subprocess.run([exe, "-C", os.path.join(PLUGDIR, "conf", "ComParE_2016_func_hip.conf"), "-I", os.path.join(G, "u3_4000.wav"),
                            "-htkoutput", htk13, "-featureSet", "is13_compare_func", "-l", "1"],
                           cwd=PLUGDIR, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and os.path.exists(htk13), r.stderr[-2000:]
        x13 = oracle.read_htk(htk13)[0]
        assert x13.shape == (1, 6373)
        assert not np.array_equal(x13, x), "featureSet = is13_compare_func gave the ComParE_2016 vector"
        assert np.isfinite(x13).mean() > 0.999


def test_plugin_mfcc_e_z_config_all_overrides(oracle):
    """config/mfcc/MFCC12_E_D_A_Z.conf, unmodified, every override active (cEnergy's HTK log branch on the raw frames,
    the reference's own cFullinputMean / cVectorConcat around the HIP components)."""
    import numpy as _np
    g = _np.load(os.path.join(ROOT, "tests", "golden", "htk_variants_synth.npz"))
    from test_oracle_pin_variants import variant_tolerance
    pcm = g["pcm_u2_16000"]
    y, tr = _run(oracle, pcm, None, "mfcc/MFCC12_E_D_A_Z.conf")
    ref = g["MFCC12_E_D_A_Z_u2_16000"]
    for comp in ("cVectorPreemphasis", "cWindower", "cTransformFFT", "cFFTmagphase", "cMelspec", "cMfcc", "cEnergy", "cDeltaRegression"):
        assert tr.get(comp, 0) > 0, f"{comp} not routed through the plugin: {tr}"
    oracle.use_reference_fft(False)
    variant_tolerance(y, ref, 12, "plugin MFCC12_E_D_A_Z", oracle.htk_variant_chain("MFCC12_E_D_A", pcm))


def test_plugin_egemaps_whole_graph(oracle):
    """The UNMODIFIED eGeMAPSv02.conf (BASELINE config 5) inside the unmodified binary with every override active: the
    GeMAPS-specific components (cSpectral's log-spectrum option sets, cSpecResample, cLpc, cFormantLpc, cHarmonics) push every
    frame through the HIP operators -- none of them falls through to the CPU code -- and the 88 functionals land within the
    chain's bars of the plain binary's (tests/test_gpu_egemaps.py explains the formant columns' floor)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "egemaps_lld_synth.npz"))
    for k in ("u2_16000", "u4_9000"):
        y, tr = _run(oracle, g["pcm_" + k], None, "egemaps/v02/eGeMAPSv02.conf", "-htkoutput")
        ref = g["func_" + k].reshape(-1, 88)
        assert y.shape == ref.shape == (1, 88)
        T20 = g["loudness_" + k].shape[0]
        T60 = g["pitch_" + k].shape[0]
        assert tr["cSpecResample"] == T20 and tr["cLpc"] == T20 and tr["cFormantLpc"] == T20, tr
        assert tr["cSpectral"] == 2 * T20 and tr["cHarmonics"] == T60, tr          # two cSpectral instances
        for comp in ("cSpectral", "cSpecResample", "cLpc", "cFormantLpc", "cHarmonics", "cTransformFFT", "cFFTmagphase", "cMelspec",
                     "cMfcc", "cPlp", "cEnergy", "cFunctionals", "cSpecScale", "cPitchShs", "cWindower"):
            assert tr.get(comp + ".cpu", 0) == 0, (comp, tr)
        # round 3: noZeroSma / onlyInSegments are per-component operators too (smilehip_window_op_row_ex): nothing runs the CPU code
        assert not [n for n, v in tr.items() if n.endswith(".cpu") and v], tr
        assert np.array_equal(y.view(np.uint32), ref.view(np.uint32)), (k, np.abs(y[0].astype(np.float64) - ref[0]).max())


def test_plugin_option_set_not_built_is_an_error(oracle, golden_synth):
    """An option set the HIP path does not cover is an ERROR of the component, named in the log -- the shipped library has no CPU
    fall-through at all (round 6: the development switch is a compile-time flag, -DSMILEHIP_DEV_CPU_FALLTHROUGH; the environment
    variable of earlier rounds does nothing): config/mfcc/MFCC12_0_D_A.conf with cMfcc doLog = 0 is not built."""
    import tempfile
    exe = os.path.join(oracle.REF_DIR, "SMILExtract")
    plug = os.path.join(PLUGDIR, "plugins", "libsmilehip_plugin.so")
    if not (os.path.exists(exe) and os.path.exists(plug)):
        pytest.skip("oracle/_ref/SMILExtract or the plugin .so not built")
    assert b"SMILEHIP_PLUGIN_ALLOW_CPU" not in open(plug, "rb").read()
    with tempfile.TemporaryDirectory() as td:
        conf = os.path.join(td, "m.conf")
        base = os.path.join(oracle.REF_DIR, "config", "mfcc", "MFCC12_0_D_A.conf")
        txt = open(base).read().replace("[mfcc:cMfcc]", "[mfcc:cMfcc]\ndoLog = 0")
        txt = txt.replace("\\{../shared/", "\\{" + os.path.join(oracle.REF_DIR, "config", "shared") + "/")
        open(conf, "w").write(txt)
        wav, out, trace = (os.path.join(td, n) for n in ("in.wav", "out.htk", "trace.txt"))
        oracle.write_wav(wav, golden_synth["pcm_u2_16000"])
        env = dict(os.environ)
        env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "opensmile_amd"), oracle.REF_DIR, env.get("LD_LIBRARY_PATH", "")])
        env["SMILEHIP_PLUGIN_TRACE"] = trace
        for extra in ({}, {"SMILEHIP_PLUGIN_ALLOW_CPU": "1"}):
            e = dict(env)
            e.update(extra)
            r = subprocess.run([exe, "-C", conf, "-I", wav, "-O", out, "-l", "2"], cwd=PLUGDIR, env=e, capture_output=True, text=True,
                               timeout=300)
            assert r.returncode != 0 and "not built for the HIP path" in (r.stderr + r.stdout), (r.returncode, (r.stderr + r.stdout)[-1500:])


def test_plugin_fused_mode_unmodified_confs(oracle, golden_synth):
    """SMILEHIP_PLUGIN_FUSE=1 with UNMODIFIED files of config/mfcc and config/plp: the plugin recognises the graph the loader's configuration
    manager holds, takes the samples the reference's wave source writes from its level, runs the whole input through the fused
    kernels in one batch, and the chain's last components hand out its rows through their own levels. Same file as the plain CPU binary within the chain's tolerance."""
    pcm = golden_synth["pcm_u10_16000"]
    for conf in ("mfcc/MFCC12_0_D_A.conf", "mfcc/MFCC12_E_D_A_Z.conf", "plp/PLP_0_D_A.conf", "plp/PLP_E_D_A.conf"):
        ref, _ = _run(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "none"}, conf)
        y, tr = _run(oracle, pcm, {"SMILEHIP_PLUGIN_FUSE": "1"}, conf)
        assert y.shape == ref.shape
        n_static = ref.shape[1] // 3
        scale = np.abs(ref[:, :n_static]).max(axis=1, keepdims=True)
        assert (np.abs(y - ref) / scale).max() <= 1e-5, conf
        frames = ref.shape[0]
        assert tr["fused.batch_frames"] == frames
        # round 5: the chain's last components write their rows at the tick level and the wave source idles -- no stage ever ticks
        assert tr["fused.rows"] >= frames and tr["fused.stage_frames"] == 0
        # ... and the block-per-tick path (round 6: every component keeps its own level and moves a block of frames per tick, on the
        # reference-order kernels): the same file within the chain's tolerance, no batch
        y3, tr3 = _run(oracle, pcm, {"SMILEHIP_PLUGIN_FUSE": "0"}, conf)
        assert tr3["fused.batch_frames"] == 0 and tr3["block.ticks"] > 0 and tr3["cFramer"] == frames, tr3
        assert (np.abs(y3 - ref) / scale).max() <= 1e-5, conf
        # no per-frame kernel launches in the chain: the stage counters stay at zero
        # (a file whose sinks do not read one cVectorConcat level -- the _Z files: cFullinputMean sits in between -- gets the static block
        # from the batch; its regression stages then run as block operators on the handed-out rows)
        for comp in ("cVectorPreemphasis", "cWindower", "cTransformFFT", "cFFTmagphase", "cMelspec", "cMfcc", "cPlp", "cEnergy"):
            assert tr.get(comp, 0) == 0, (conf, comp, tr)
    # a graph that is not a cepstral chain stays on the per-component path, and says so
    y, tr = _run(oracle, golden_synth["pcm_u3_16000"], {"SMILEHIP_PLUGIN_FUSE": "1", "SMILEHIP_PLUGIN_COMPONENTS": "cMelspec,cMfcc"},
                 "is09-13/IS09_emotion.conf", "-lldhtkoutput")
    assert tr["fused.batch_frames"] == 0 and tr["cMfcc"] > 0


def test_plugin_fused_mode_is_the_default(oracle, golden_synth, monkeypatch):
    """Round 5: with SMILEHIP_PLUGIN_FUSE unset the unmodified MFCC12_0_D_A.conf runs fused (one batch per file, no per-frame
    device round trip), quietly; SMILEHIP_PLUGIN_FUSE=0 keeps the components apart; a graph the reader does not recognise
    stays on the per-component path without a warning."""
    pcm = golden_synth["pcm_u10_16000"]
    ref, _ = _run(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "none"})
    monkeypatch.delenv("SMILEHIP_PLUGIN_FUSE", raising=False)
    y, tr = _run(oracle, pcm)
    assert tr["fused.batch_frames"] == ref.shape[0] and tr.get("cMfcc", 0) == 0 and tr.get("cTransformFFT", 0) == 0
    scale = np.abs(ref[:, :13]).max(axis=1, keepdims=True)
    assert (np.abs(y - ref) / scale).max() <= 1e-5
    y0, tr0 = _run(oracle, pcm, {"SMILEHIP_PLUGIN_FUSE": "0"})
    assert tr0["fused.batch_frames"] == 0 and tr0["cMfcc"] == ref.shape[0]


def test_plugin_fused_tick_level_text_sink_time_stamps(oracle):
    """The CSV sink prints every row's frameTime: the rows the fused chain's last components write at the tick level carry the
    time stamps the framer would have given them -- the file equals the per-frame hand-out's and differs from the CPU binary's
    only in the values (the fast kernel's 1e-5), never in a name, a row count or a time stamp."""
    from opensmile_amd import synth
    pcm = synth.utterance(33, 20000)
    for conf in ("mfcc/MFCC12_0_D_A.conf",):              # (the one cepstral file with a text sink: standard_data_output_lldonly.conf.inc)
        ref, _ = _run_bytes(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "none"}, conf, "-csvoutput")
        tick, tr = _run_bytes(oracle, pcm, {"SMILEHIP_PLUGIN_FUSE": "1"}, conf, "-csvoutput")
        per_frame, _ = _run_bytes(oracle, pcm, {"SMILEHIP_PLUGIN_FUSE": "1", "SMILEHIP_PLUGIN_FUSE_TICK": "0"}, conf, "-csvoutput")
        assert tr["fused.stage_frames"] == 0 and tr["fused.batch_frames"] > 0
        static, trs = _run_bytes(oracle, pcm, {"SMILEHIP_PLUGIN_FUSE": "1", "SMILEHIP_PLUGIN_FUSE_TICK": "static"}, conf, "-csvoutput")
        assert trs["fused.stage_frames"] == 0 and static == per_frame, conf     # (the static block at the tick level, the reference's deltas)
        assert tick == per_frame, conf                     # (the batch's own regression stages: the same bits)
        rl, tl = ref.decode().splitlines(), tick.decode().splitlines()
        assert len(rl) == len(tl) and rl[0] == tl[0], conf
        for a, b in zip(rl[1:], tl[1:]):
            fa, fb = a.split(";"), b.split(";")
            assert fa[:2] == fb[:2] and len(fa) == len(fb), (conf, a[:60], b[:60])      # name and frameTime


def test_plugin_viterbi_tick_level_override(oracle, golden_f0):
    """cPitchSmootherViterbi, cValbasedSelector and cPitchJitter as tick-level overrides (myTick replaced; the selectors decide
    per frame whether it is handed on, zeroed or dropped; the jitter component reads the wave stretch its carried state asks
    for and hands it to the device-resident stream). cPitchSmootherViterbi (myTick replaced: one frame of candidates per tick goes to the
    device-resident trellis, decided frames are written at the tick at which the reference's incremental scheme releases
    them). With ONLY this component overridden, the unmodified ComParE_2016.conf and eGeMAPSv02.conf (bufferLength 40,
    F0finalLog) must reproduce the plain binary's LLD files BIT FOR BIT -- everything downstream (energy gate, jitter,
    smoothers with their end-of-input behaviour) depends on which frames appear at which tick."""
    from test_oracle_pin_f0 import KEYS
    for k in KEYS[:3]:
        pcm = golden_f0["pcm_" + k]
        for conf in (COMPARE, "egemaps/v02/eGeMAPSv02.conf"):
            ref, _ = _run(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "none"}, conf, "-lldhtkoutput")
            y, tr = _run(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "cPitchSmootherViterbi,cValbasedSelector,cPitchJitter"}, conf, "-lldhtkoutput")
            assert tr["cPitchJitter"] > 0 and tr["cPitchJitter.cpu"] == 0
            assert tr["cPitchSmootherViterbi"] > 0 and tr["cPitchSmootherViterbi.cpu"] == 0
            assert tr["cValbasedSelector"] > 0 and tr["cValbasedSelector.cpu"] == 0
            assert y.shape == ref.shape
            assert np.array_equal(y.view(np.uint32), ref.view(np.uint32)), (k, conf, int((y != ref).any(axis=1).sum()))


def test_plugin_compare16_every_component_overridden(oracle, golden_f0):
    """The unmodified ComParE_2016.conf with EVERY overridable component on HIP -- the 20 ms and 60 ms front ends, cSpecScale,
    cPitchShs, the tick-level cPitchSmootherViterbi / cValbasedSelector / cPitchJitter, cSpectral, cPlp, the smoothers and
    deltas (noZeroSma and onlyInSegments variants included): the binary's 130-column LLD file bit for bit, no frame on the CPU code."""
    from test_gpu_compare_full import AB, F0, KEYS_130, f0_lld_tolerances
    from test_oracle_pin_compare import compare_tolerances
    for k in KEYS_130[:2]:
        y, tr = _run(oracle, golden_f0["pcm_" + k], {}, COMPARE, "-lldhtkoutput")   # no selection: every override
        ref = golden_f0["lld130_" + k]
        assert y.shape == ref.shape
        compare_tolerances(y[:, AB], ref[:, AB], "plugin " + k)
        f0_lld_tolerances(y[:, F0], ref[:, F0], "plugin " + k)
        for comp in ("cTransformFFT", "cSpecScale", "cPitchShs", "cPitchSmootherViterbi", "cValbasedSelector", "cPitchJitter", "cSpectral", "cPlp",
                     "cContourSmoother", "cDeltaRegression", "cMelspec", "cMfcc", "cEnergy", "cMZcr"):
            assert tr.get(comp, 0) > 0, (comp, tr)
        # round 3: the F0 group's noZeroSma smoother and onlyInSegments delta are HIP operators as well: no CPU fall-through
        assert not [n for n, v in tr.items() if n.endswith(".cpu") and v], tr


@pytest.mark.gpu
def test_plugin_fused_source_big_set_levels(oracle):
    """cHipLldSource for the three big sets: featureSet = <set>_lld puts the whole LLD level (rows, end-of-input rows, time
    stamps, names) into the level `lld`, <set>_func the functionals vector into `func`; the reference's own sinks write the
    files. They must be the files smilextract_hip writes for the same input (same device path, same numbers: byte for byte),
    which tests/test_host_io.py holds against the real binary's files; the CSV head line is compared with the binary's here too."""
    exe = os.path.join(oracle.REF_DIR, "SMILExtract")
    plug = os.path.join(PLUGDIR, "plugins", "libsmilehip_plugin.so")
    hipexe = os.path.join(ROOT, "opensmile_amd", "smilextract_hip")
    if not (os.path.exists(exe) and os.path.exists(plug) and os.path.exists(hipexe)):
        pytest.skip("oracle/_ref/SMILExtract, the plugin .so or smilextract_hip not built")
    G = os.path.join(ROOT, "tests", "golden", "files")
    wav = os.path.join(G, "u3_4000.wav")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "opensmile_amd"), oracle.REF_DIR, env.get("LD_LIBRARY_PATH", "")])
    with tempfile.TemporaryDirectory() as td:
        for fset, hipset, gold in (("is09", "is09_emotion", "is09_lld_u3.csv"), ("compare16", "compare16", "compare16_lld_u3.csv"),
                                   ("is13_compare", "is13_compare", None), ("egemapsv02", "egemapsv02", "egemaps_lld_u3.csv")):
            p_csv, p_htk, p_fhtk = (os.path.join(td, f"p_{fset}{e}") for e in (".csv", ".htk", ".f.htk"))
            h_csv, h_htk, h_fhtk = (os.path.join(td, f"h_{fset}{e}") for e in (".csv", ".htk", ".f.htk"))
            r = subprocess.run([exe, "-C", os.path.join(PLUGDIR, "conf", "LLD_hip.conf"), "-featureSet", fset + "_lld", "-I", wav,
                                "-lldcsvoutput", p_csv, "-lldhtkoutput", p_htk, "-instname", "u3", "-l", "1"],
                               cwd=PLUGDIR, env=env, capture_output=True, text=True, timeout=300)
            assert r.returncode == 0 and os.path.exists(p_csv) and os.path.exists(p_htk), (fset, r.stderr[-2000:])
            r = subprocess.run([exe, "-C", os.path.join(PLUGDIR, "conf", "ComParE_2016_func_hip.conf"), "-featureSet", fset + "_func", "-I", wav,
                                "-htkoutput", p_fhtk, "-instname", "u3", "-l", "1"],
                               cwd=PLUGDIR, env=env, capture_output=True, text=True, timeout=300)
            assert r.returncode == 0 and os.path.exists(p_fhtk), (fset, r.stderr[-2000:])
            subprocess.run([hipexe, "--set", hipset, "-I", wav, "-lldcsvoutput", h_csv, "-lldhtkoutput", h_htk, "-htkoutput", h_fhtk,
                            "-instname", "u3"], check=True, env=env)
            assert open(p_htk, "rb").read() == open(h_htk, "rb").read(), fset
            assert open(p_csv).read() == open(h_csv).read(), fset
            assert open(p_fhtk, "rb").read() == open(h_fhtk, "rb").read(), fset
            if gold:
                assert open(p_csv).readline() == open(os.path.join(G, gold)).readline(), fset


def test_plugin_option_sets(oracle):
    """The components' other option sets (VERDICT r2 missing 3) through the plugin: cTransformFFT inverse = 1, cFFTmagphase normalise /
    power / dBpsd / phase / joinMagphase, cMZcr mcr / amax / maxmin / dc, cPitchACF's HNR / voiceQual outputs, cPlp RASTA = 1, cMelspec on other
    spectral scales and with the erb (HFCC) / custom bandwidth banks (tests/conf/option_sets.conf). The same binary with and
    without the overrides: every value the reference's bits (log10f and atan2f follow glibc's algorithms, glibc_float.hpp)."""
    from opensmile_amd import synth
    conf = os.path.join(ROOT, "tests", "conf", "option_sets.conf")
    pcm = synth.utterance(5, 12000)
    pcm[2000:2400] = 0                                    # a stretch of exact zeros (zero-crossing rules, dBpsd floor, phase of 0)
    ref, tr0 = _run(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "none"}, conf)
    y, tr = _run(oracle, pcm, None, conf)
    assert not any(tr0.values()) and ref.shape == y.shape == (ref.shape[0], 2441)
    T = ref.shape[0]
    assert tr.get("cTransformFFT", 0) == 2 * T and tr.get("cFFTmagphase", 0) == 7 * T and tr.get("cMZcr", 0) == T, tr
    assert tr.get("cPitchACF", 0) == T and tr.get("cAcf", 0) == 2 * T and tr.get("cPlp", 0) == T and tr.get("cMelspec", 0) == 5 * T, tr
    assert not any(k.endswith(".cpu") and v for k, v in tr.items()), tr
    cols = {"ifft": (0, 512), "magN": (512, 769), "magNP": (769, 1026), "magP": (1026, 1283), "magDb": (1283, 1540),
            "join_mag": (1540, 1797), "join_phase": (1797, 2054), "phase": (2054, 2311), "mzcr": (2311, 2317),
            "pitch [voiceProb HNR HNRdB linHNR voiceQual F0 F0raw F0env]": (2317, 2325), "cPlp RASTA = 1": (2325, 2351),
            "cMelspec bark": (2351, 2375), "cMelspec bwMethod erb (HFCC)": (2375, 2395), "cMelspec semitone": (2395, 2425),
            "cMelspec bark_schroed, custom bandwidth": (2425, 2441)}
    for name, (a, b) in cols.items():
        g, r = np.ascontiguousarray(y[:, a:b]), np.ascontiguousarray(ref[:, a:b])
        d = g.view(np.uint32) != r.view(np.uint32)
        assert not d.any(), f"{name}: {d.sum()} of {d.size} values differ, first at {np.argwhere(d)[0]}: {g[d][:3]} vs {r[d][:3]}"


@pytest.mark.parametrize("fs", [8000, 44100])
@pytest.mark.parametrize("conf,out_opt", [("egemaps/v02/eGeMAPSv02.conf", "-lldhtkoutput"), ("compare16/ComParE_2016.conf", "-lldhtkoutput")])
def test_plugin_big_sets_at_other_sample_rates(oracle, conf, out_opt, fs):
    """The unmodified big-set files inside the unmodified binary on 8 kHz / 44.1 kHz input with every override active: the
    per-component operators take the spectrum sizes of that rate (cSpectral on 129 / 513 bins, cSpecScale / cPitchShs / cHarmonics on
    FFT 512 / 4096, cSpecResample from FFT 256 / 1024): the plain binary's LLD file bit for bit, nothing on the CPU code."""
    from opensmile_amd import synth
    pcm = synth.utterance(33, int(0.8 * fs), fs)
    ref, _ = _run(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "none"}, conf, out_opt, fs)
    y, tr = _run(oracle, pcm, None, conf, out_opt, fs)
    assert y.shape == ref.shape
    assert not [n for n, v in tr.items() if n.endswith(".cpu") and v], tr
    d = y.view(np.uint32) != ref.view(np.uint32)
    assert not d.any(), f"{d.sum()} of {d.size} cells differ, columns {sorted(set(np.argwhere(d)[:, 1]))[:20]}"


@pytest.mark.parametrize("conf,n_func", [("is09-13/IS09_emotion.conf", 384), ("compare16/ComParE_2016.conf", 6373),
                                         ("is09-13/IS13_ComParE.conf", 6373), ("egemaps/v02/eGeMAPSv02.conf", 88)])
def test_plugin_fused_mode_unmodified_big_sets(oracle, conf, n_func):
    """SMILEHIP_PLUGIN_FUSE=1 with the UNMODIFIED big-set files (VERDICT r2 missing 4): one fused batch computes the set's LLD level for
    the whole file, the final smoother / delta instances hand out its rows, everything upstream writes zeros, the cFunctionals
    overrides run on the handed-out levels: LLD file and functionals file equal the plain binary's bit for bit."""
    from opensmile_amd import synth
    pcm = synth.utterance(71, 40000)
    outs = {}
    for tag, env in (("cpu", {"SMILEHIP_PLUGIN_COMPONENTS": "none"}), ("fused", {"SMILEHIP_PLUGIN_FUSE": "1"})):
        lld, tr = _run(oracle, pcm, env, conf, "-lldhtkoutput")
        fn, tr2 = _run(oracle, pcm, env, conf, "-htkoutput")
        outs[tag] = (lld, fn, tr, tr2)
    lld_c, fn_c = outs["cpu"][:2]
    lld_f, fn_f, tr, tr2 = outs["fused"]
    assert tr.get("fused.rows", 0) > 0 and tr.get("fused.batch_frames", 0) == lld_c.shape[0], tr
    assert not [n for n, v in tr.items() if n.endswith(".cpu") and v], tr
    assert lld_f.shape == lld_c.shape and np.array_equal(lld_f.view(np.uint32), lld_c.view(np.uint32)), \
        sorted(set(np.argwhere(lld_f.view(np.uint32) != lld_c.view(np.uint32))[:, 1]))[:20]
    assert fn_f.shape == fn_c.shape == (1, n_func)
    d = fn_f.view(np.uint32) != fn_c.view(np.uint32)
    assert not d.any(), f"{d.sum()} of {d.size} functionals differ, first {np.argwhere(d)[:10, 1]}"
    # no per-frame device work upstream of the fused levels
    for comp in ("cTransformFFT", "cSpecScale", "cPitchShs", "cSpectral", "cHarmonics", "cAcf"):
        assert tr.get(comp, 0) == 0, (comp, tr)


def test_plugin_is10_paraling(oracle):
    """The UNMODIFIED config/is09-13/IS10_paraling.conf (the INTERSPEECH 2010 Paralinguistic Challenge set: 38 LLDs + deltas, 1582
    functionals) inside the unmodified binary with every override active: the components the five BASELINE configs do not have --
    cIntensity, cLsp, cPitchSmoother, cVectorOperation, cSpecResample / cLpc on 25 ms frames (512 -> 275 samples, p = 8), cSpecScale with
    minF = 20, cPitchShs without F0raw / voicingClip, the Onset functionals -- run on the GPU: LLD file and functionals file equal the
    plain binary's bit for bit, nothing on the reference's CPU code."""
    from opensmile_amd import synth
    conf = "is09-13/IS10_paraling.conf"
    for u, n in ((71, 24000), (5, 9000)):
        pcm = synth.utterance(u, n)
        ref_l, _ = _run(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "none"}, conf, "-lldhtkoutput")
        ref_f, _ = _run(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "none"}, conf, "-htkoutput")
        y_l, tr = _run(oracle, pcm, None, conf, "-lldhtkoutput")
        y_f, tr2 = _run(oracle, pcm, None, conf, "-htkoutput")
        assert not [k for k, v in tr.items() if k.endswith(".cpu") and v], tr
        T25 = 1 + (n - 400) // 160
        for comp, cnt in (("cIntensity", T25), ("cLsp", T25), ("cVectorOperation", T25), ("cSpecResample", T25), ("cLpc", T25)):
            assert tr.get(comp, 0) == cnt, (comp, tr)
        assert tr.get("cPitchSmoother", 0) > 0 and tr.get("cPitchShs", 0) > 0 and tr.get("cSpecScale", 0) > 0 and tr.get("cPitchJitter", 0) > 0, tr
        assert tr.get("cFunctionals", 0) > 0, tr
        assert ref_l.shape == y_l.shape and ref_l.shape[1] == 76
        d = y_l.view(np.uint32) != ref_l.view(np.uint32)
        assert not d.any(), f"LLD: {d.sum()} of {d.size} cells differ, columns {sorted(set(np.argwhere(d)[:, 1]))[:20]}"
        assert ref_f.shape == y_f.shape == (1, 1582)
        d = y_f.view(np.uint32) != ref_f.view(np.uint32)
        assert not d.any(), f"functionals: {d.sum()} of {d.size} differ at {np.argwhere(d)[:10, 1]}: {y_f[d][:5]} vs {ref_f[d][:5]}"


@pytest.mark.parametrize("conf,n_lld,n_func", [("is09-13/IS11_speaker_state.conf", 118, 4368),
                                               ("is09-13/IS12_speaker_trait.conf", 120, 5757), ("is09-13/IS12_speaker_trait_compat.conf", 128, 6125)])
def test_plugin_other_interspeech_sets(oracle, conf, n_lld, n_func):
    """The other INTERSPEECH challenge sets of config/is09-13, unmodified, with every override active: every component and every
    functional family they use is an operator of the library (IS11 / IS12: cPitchShs with four candidates and the older peak picker
    (greedyPeakAlgo = 0), cSpectral without centroid / with the band 25-650, the vector sums of cVectorOperation; IS11: the older Peaks
    functionals; IS12: Peaks2 / Segments / Lpc and the Viterbi smoother on five states) -- nothing runs the reference's CPU code, and the
    LLD and functionals files equal the plain binary's bit for bit."""
    from opensmile_amd import synth
    pcm = synth.utterance(71, 24000)
    ref_l, _ = _run(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "none"}, conf, "-lldhtkoutput")
    ref_f, _ = _run(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "none"}, conf, "-htkoutput")
    y_l, tr = _run(oracle, pcm, None, conf, "-lldhtkoutput")
    y_f, tr2 = _run(oracle, pcm, None, conf, "-htkoutput")
    assert not [k for k, v in tr.items() if k.endswith(".cpu") and v], tr
    assert tr.get("cFunctionals", 0) > 0 and tr.get("cPitchShs", 0) > 0, tr
    assert ref_l.shape == y_l.shape and ref_l.shape[1] == n_lld and ref_f.shape == y_f.shape == (1, n_func), (ref_l.shape, y_l.shape, ref_f.shape, y_f.shape)
    d = y_l.view(np.uint32) != ref_l.view(np.uint32)
    assert not d.any(), f"LLD: {d.sum()} of {d.size} cells differ, columns {sorted(set(np.argwhere(d)[:, 1]))[:20]}"
    d = y_f.view(np.uint32) != ref_f.view(np.uint32)
    assert not d.any(), f"functionals: {d.sum()} of {d.size} differ at {np.argwhere(d)[:10, 1]}: {y_f[d][:5]} vs {ref_f[d][:5]}"


def _run_taps(oracle, pcm, env_extra, conf, expect_fail=None):
    """a run of a tests/conf file whose only outputs are its HTK taps (-T): {tap name: file bytes}, the plugin's frame counters"""
    exe = os.path.join(oracle.REF_DIR, "SMILExtract")
    plug = os.path.join(PLUGDIR, "plugins", "libsmilehip_plugin.so")
    if not (os.path.exists(exe) and os.path.exists(plug)):
        pytest.skip("oracle/_ref/SMILExtract or the plugin .so not built (needs /root/reference at build time)")
    with tempfile.TemporaryDirectory() as td:
        wav, trace = os.path.join(td, "in.wav"), os.path.join(td, "trace.txt")
        oracle.write_wav(wav, pcm, 16000)
        env = dict(os.environ)
        env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "opensmile_amd"), oracle.REF_DIR, env.get("LD_LIBRARY_PATH", "")])
        env["SMILEHIP_PLUGIN_TRACE"] = trace
        env.update(env_extra or {})
        r = subprocess.run([exe, "-C", conf, "-I", wav, "-T", td, "-l", "1"], cwd=PLUGDIR, env=env, capture_output=True, text=True,
                           errors="replace", timeout=300)
        if expect_fail is not None:
            assert r.returncode != 0 and expect_fail in r.stderr + r.stdout, (r.returncode, r.stderr[-2000:])
            return None, None
        assert r.returncode == 0, r.stderr[-2000:]
        taps = {n[4:-4]: open(os.path.join(td, n), "rb").read() for n in sorted(os.listdir(td)) if n.startswith("tap_")}
        tr = dict(l.split() for l in open(trace).read().split("\n") if l.strip()) if os.path.exists(trace) else {}
    return taps, {k: int(v) for k, v in tr.items()}


def _segments_pcm():
    from opensmile_amd import synth
    pcm = synth.utterance(9, 48000).copy()
    for a, b in ((8000, 12000), (23700, 24200), (30000, 36000)):
        pcm[a:b] = 0                                     # RMS energy exactly 0, log energy at its floor: plateaus for chX
    return pcm


def test_plugin_segments_every_algorithm(oracle):
    """tests/conf/segments_family.conf: fifteen cFunctionals instances, one per segmentationAlgorithm / option variant of
    cFunctionalSegments (delta, delt2, (m)(NA)relTh, (NA)absTh, chX, the names that fall back to delta), behind the plugin: every tap
    the plain binary's bytes, nothing on the CPU."""
    conf = os.path.join(ROOT, "tests", "conf", "segments_family.conf")
    pcm = _segments_pcm()
    ref, tr0 = _run_taps(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "none"}, conf)
    own, tr = _run_taps(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "cFunctionals"}, conf)
    assert not any(tr0.values()) and len(ref) == 16 and sorted(own) == sorted(ref)
    assert tr.get("cFunctionals", 0) >= 15 and not [k for k, v in tr.items() if k.endswith(".cpu") and v], tr
    for k in ref:
        assert len(ref[k]) > 12 and own[k] == ref[k], k


def test_plugin_percentile_quotients_and_level_times(oracle):
    """tests/conf/quotients_leveltimes.conf: cFunctionals instances with Percentiles.pctlquotient[] (with and without pctlrange[])
    and Times.upleveltime[] / downleveltime[] behind the plugin: every tap the plain binary's bytes, nothing on the CPU."""
    conf = os.path.join(ROOT, "tests", "conf", "quotients_leveltimes.conf")
    pcm = _segments_pcm()
    ref, tr0 = _run_taps(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "none"}, conf)
    own, tr = _run_taps(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "cFunctionals"}, conf)
    assert not any(tr0.values()) and len(ref) == 7 and sorted(own) == sorted(ref)
    assert tr.get("cFunctionals", 0) >= 6 and not [k for k, v in tr.items() if k.endswith(".cpu") and v], tr
    for k in ref:
        assert len(ref[k]) > 12 and own[k] == ref[k], k


def test_plugin_modulation_family(oracle):
    """tests/conf/modulation_family.conf: three cFunctionals instances with the Modulation family (default options; frame-based
    windows, Hann, its own axis; removeNonZeroMean, rectangle) behind the plugin: every tap the plain binary's bytes, nothing on
    the CPU."""
    conf = os.path.join(ROOT, "tests", "conf", "modulation_family.conf")
    from opensmile_amd import synth
    for pcm in (_segments_pcm(), synth.utterance(71, 160000)):
        ref, tr0 = _run_taps(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "none"}, conf)
        own, tr = _run_taps(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "cFunctionals"}, conf)
        assert not any(tr0.values()) and len(ref) == 4 and sorted(own) == sorted(ref)
        assert tr.get("cFunctionals", 0) >= 3 and not [k for k, v in tr.items() if k.endswith(".cpu") and v], tr
        for k in ref:
            assert len(ref[k]) > 12 and own[k] == ref[k], k


def test_plugin_refuses_what_is_not_built(oracle):
    """A cFunctionals instance with an option that is not an operator of the library (Segments.growDynSegBuffer: the segment buffer
    that grows past maxNumSeg): the override says so and the process fails -- no CPU path, silent or otherwise, in the shipped library
    (the environment variable of earlier rounds changes nothing)."""
    pcm = _segments_pcm()
    with tempfile.TemporaryDirectory() as td:
        conf = os.path.join(td, "grow.conf")
        text = open(os.path.join(ROOT, "tests", "conf", "segments_family.conf")).read()
        key = "Segments.segmentationAlgorithm = delta\n"
        assert text.count(key) == 2
        open(conf, "w").write(text.replace(key, key + "Segments.growDynSegBuffer = 1\n", 1))
        _run_taps(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "cFunctionals"}, conf,
                  expect_fail="cFunctionals: a functional family or option of this instance is not built")
        _run_taps(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "cFunctionals", "SMILEHIP_PLUGIN_ALLOW_CPU": "1"}, conf,
                  expect_fail="cFunctionals: a functional family or option of this instance is not built")


@pytest.mark.parametrize("conf", ["avec11-14/avec2011.conf", "avec11-14/avec2013.conf"])
def test_plugin_avec_sets_whole(oracle, conf):
    """avec2011.conf / avec2013.conf, unmodified, every override active: their functionals use Segments' NArelTh and chX algorithms
    (functionalSegments.cpp:369-413, :560-653) next to nonX / eqX. Nothing on the CPU; the plain binary's file byte for byte."""
    from opensmile_amd import synth
    pcm = synth.utterance(71, 24000)
    ref, _ = _run_bytes(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "none"}, conf, "-O")
    y, tr = _run_bytes(oracle, pcm, None, conf, "-O")
    assert not [k for k, v in tr.items() if k.endswith(".cpu") and v], tr
    assert tr.get("cFunctionals", 0) > 0 and tr.get("cSpectral", 0) > 0, tr
    assert len(ref) > 100 and y == ref


def test_plugin_harmonics_acf_hnr_alone(oracle):
    """prosodyShsViterbiLoudness.conf asks cHarmonics for the ACF harmonics-to-noise ratio alone (computeAcfHnrLogdB on a 55 ms frame's
    1024-point spectrum): column 0 of the GeMAPS operator's row. Every override active, nothing on the CPU, byte-identical file."""
    from opensmile_amd import synth
    pcm = synth.utterance(71, 24000)
    conf = "prosody/prosodyShsViterbiLoudness.conf"
    ref, _ = _run_bytes(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "none"}, conf, "-O")
    y, tr = _run_bytes(oracle, pcm, None, conf, "-O")
    assert not [k for k, v in tr.items() if k.endswith(".cpu") and v], tr
    assert tr.get("cHarmonics", 0) > 0 and tr.get("cPitchShs", 0) > 0, tr
    assert len(ref) > 100 and y == ref


@pytest.mark.parametrize("conf,n_lld", [("is09-13/IS10_paraling_compat.conf", 76), ("emobase/emobase2010.conf", None)])
def test_plugin_specscale_switch_sets(oracle, conf, n_lld):
    """IS10_paraling_compat.conf and emobase2010.conf run cSpecScale on the log2 axis WITHOUT smoothing / enhancement / auditory
    weighting (smilehip_lld_config::specscale_off): every override active, nothing on the CPU, the plain binary's file byte for byte."""
    from opensmile_amd import synth
    pcm = synth.utterance(71, 24000)
    opt = "-lldhtkoutput" if n_lld else "-O"
    ref, _ = _run_bytes(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "none"}, conf, opt)
    y, tr = _run_bytes(oracle, pcm, None, conf, opt)
    assert not [k for k, v in tr.items() if k.endswith(".cpu") and v], tr
    assert tr.get("cSpecScale", 0) > 0, tr
    assert len(ref) > 100 and y == ref


@pytest.mark.parametrize("fs", [8000, 44100])
def test_plugin_is10_paraling_other_rates(oracle, fs):
    """IS10_paraling at 8 kHz and 44.1 kHz: the any-geometry operators (cSpecResample from 256 / 2048 spectrum values, cLpc on the
    resampled frame, cSpecScale / cPitchShs on 512- / 4096-point transforms, the jitter pass at that rate): bit for bit, nothing on the CPU."""
    from opensmile_amd import synth
    conf = "is09-13/IS10_paraling.conf"
    pcm = synth.utterance(33, int(0.8 * fs), fs)
    ref, _ = _run(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "none"}, conf, "-lldhtkoutput", fs)
    y, tr = _run(oracle, pcm, None, conf, "-lldhtkoutput", fs)
    assert not [k for k, v in tr.items() if k.endswith(".cpu") and v], tr
    assert y.shape == ref.shape and ref.shape[1] == 76
    d = y.view(np.uint32) != ref.view(np.uint32)
    assert not d.any(), f"{d.sum()} of {d.size} cells differ, columns {sorted(set(np.argwhere(d)[:, 1]))[:20]}"


def _run_bytes(oracle, pcm, env_extra, conf, out_opt="-O"):
    """the output file of one run as bytes (text sinks: ARFF / CSV), and the plugin's frame counters"""
    exe = os.path.join(oracle.REF_DIR, "SMILExtract")
    plug = os.path.join(PLUGDIR, "plugins", "libsmilehip_plugin.so")
    if not (os.path.exists(exe) and os.path.exists(plug)):
        pytest.skip("oracle/_ref/SMILExtract or the plugin .so not built (needs /root/reference at build time)")
    with tempfile.TemporaryDirectory() as td:
        wav, out, trace = (os.path.join(td, n) for n in ("in.wav", "out.bin", "trace.txt"))
        oracle.write_wav(wav, pcm, 16000)
        env = dict(os.environ)
        env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "opensmile_amd"), oracle.REF_DIR, env.get("LD_LIBRARY_PATH", "")])
        env["SMILEHIP_PLUGIN_TRACE"] = trace
        env.update(env_extra or {})
        r = subprocess.run([exe, "-C", os.path.join(oracle.REF_DIR, "config", conf), "-I", wav, out_opt, out, "-l", "1"],
                           cwd=PLUGDIR, env=env, capture_output=True, text=True, errors="replace", timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        data = open(out, "rb").read()
        tr = dict(l.split() for l in open(trace).read().split("\n") if l.strip()) if os.path.exists(trace) else {}
    return data, {k: int(v) for k, v in tr.items()}


@pytest.mark.parametrize("conf", ["avec11-14/avec2011.conf", "avec11-14/avec2013.conf", "misc/emo_large.conf",
                                  "mediaeval12/MediaEval_Audio_IS12based_subwin2.conf"])
def test_plugin_general_spectral_sets(oracle, conf):
    """The shipped files whose cSpectral instance asks for another descriptor set than ComParE_2016's or GeMAPS' (four bands, maxPos /
    minPos, no centroid ...): the general operator (smilehip_spectral_op_*) takes them -- the file the plugin run writes equals the
    plain binary's byte for byte, and nothing ran on the CPU."""
    from opensmile_amd import synth
    pcm = synth.utterance(71, 24000)
    ref, _ = _run_bytes(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "none"}, conf)
    # every LLD override; these files' cFunctionals instances use families that are not operators of the library (Crossings, DCT,
    # Peaks, pctlquotient ...: refused by name when overridden) and stay the reference's
    lld = ("cWindower,cVectorPreemphasis,cTransformFFT,cFFTmagphase,cMelspec,cMfcc,cEnergy,cMZcr,cAcf,cPitchACF,cDeltaRegression,"
           "cContourSmoother,cSpectral,cPlp,cSpecScale,cPitchShs,cPitchSmootherViterbi,cValbasedSelector,cPitchJitter,cVectorOperation,"
           "cIntensity,cLsp,cPitchSmoother,cSpecResample,cLpc,cFormantLpc,cHarmonics")
    own, tr = _run_bytes(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": lld}, conf)
    assert not [k for k, v in tr.items() if k.endswith(".cpu") and v], tr
    assert tr.get("cSpectral", 0) > 0, tr
    assert len(ref) > 100 and own == ref


def test_plugin_delta_variants(oracle):
    """cDeltaRegression's option variants (deltaRegression.cpp:104-170; smilehip_delta_op_row): relativeDelta, halfWaveRect, absOutput
    (halfWaveRect wins over it), deltawin = 0 (the simple difference), and their combinations with onlyInSegments, on an energy contour
    and an F0 contour with unvoiced zeros (tests/conf/delta_variants.conf). The same binary with and without the overrides: bit for
    bit, nothing on the CPU."""
    from opensmile_amd import synth
    conf = os.path.join(ROOT, "tests", "conf", "delta_variants.conf")
    pcm = np.concatenate([synth.utterance(5, 24000), np.zeros(1600, np.int16), synth.utterance(7, 8000)])
    ref, tr0 = _run(oracle, pcm, {"SMILEHIP_PLUGIN_COMPONENTS": "none"}, conf)
    y, tr = _run(oracle, pcm, None, conf)
    assert not any(tr0.values()) and ref.shape == y.shape and ref.shape[1] == 16
    assert (ref[:, 1] == 0).any() and (ref[:, 1] > 0).any()               # the F0 contour has voiced and unvoiced frames
    assert tr.get("cDeltaRegression", 0) >= 14 * (ref.shape[0] - 4), tr
    assert not any(k.endswith(".cpu") and v for k, v in tr.items()), tr
    names = ["contours", "relativeDelta", "halfWaveRect", "absOutput", "deltawin 0", "deltawin 0 + relative + segments + abs",
             "relative + segments", "deltawin 3 + relative + halfWave (+ abs)"]
    for i, name in enumerate(names):
        g, r = np.ascontiguousarray(y[:, 2 * i:2 * i + 2]), np.ascontiguousarray(ref[:, 2 * i:2 * i + 2])
        d = g.view(np.uint32) != r.view(np.uint32)
        assert not d.any(), f"{name}: {d.sum()} of {d.size} values differ, first at {np.argwhere(d)[0]}: {g[d][:3]} vs {r[d][:3]}"


@pytest.mark.parametrize("case", ["all", "noflux", "posdiff_only"])
def test_plugin_spectral_round6_options(oracle, case, tmp_path):
    """cSpectral option sets no shipped file uses (specDiff, specPosDiff, fluxCentroid, fluxAtFluxCentroid, standardDeviation, slopes[];
    round 6): a second cSpectral instance behind avec2011's magnitude level -- the plugin's output level equals the plain binary's
    byte for byte (incl. the reference's single zero for the whole flux family on the first frame), every frame through the override."""
    from test_oracle_pin_spectral_sets import EXTRA, spectral_section
    from opensmile_amd import synth
    exe = os.path.join(oracle.REF_DIR, "SMILExtract")
    plug = os.path.join(PLUGDIR, "plugins", "libsmilehip_plugin.so")
    if not (os.path.exists(exe) and os.path.exists(plug)):
        pytest.skip("oracle/_ref/SMILExtract or the plugin .so not built")
    bands, slopes, flags = EXTRA[case]
    td = str(tmp_path)
    wav = os.path.join(td, "in.wav")
    oracle.write_wav(wav, synth.utterance(9, 32000), 16000)
    outs = {}
    for mode in ("plain", "plugin"):
        c = os.path.join(td, f"{mode}.conf")
        open(c, "w").write("\\{%s}\n[componentInstances:cComponentManager]\ninstance[spec2].type=cSpectral\ninstance[tap_out].type=cHtkSink\n"
                           "%s[tap_out:cHtkSink]\nreader.dmLevel=spectral2\nfilename=%s/tap_%s.htk\n"
                           % (os.path.join(oracle.REF_DIR, "config", "avec11-14/avec2011.conf"),
                              spectral_section("spec2", "fftmagH25", "spectral2", bands, slopes, flags), td, mode))
        env = dict(os.environ)
        env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "opensmile_amd"), oracle.REF_DIR, env.get("LD_LIBRARY_PATH", "")])
        trace = os.path.join(td, "trace.txt")
        env["SMILEHIP_PLUGIN_TRACE"] = trace
        r = subprocess.run([exe, "-C", c, "-I", wav, "-O", os.path.join(td, "o.bin"), "-l", "1"], cwd=PLUGDIR if mode == "plugin" else td, env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.stderr + r.stdout)[-1500:]
        outs[mode] = open(os.path.join(td, f"tap_{mode}.htk"), "rb").read()
        if mode == "plugin":
            tr = dict(l.split() for l in open(trace) if len(l.split()) == 2)
            assert int(tr["cSpectral"]) > 0 and int(tr.get("cSpectral.cpu", 0)) == 0
    assert len(outs["plain"]) > 1000 and outs["plain"] == outs["plugin"]


@pytest.mark.parametrize("stage", [1, 2])
def test_plugin_plp_partial_modes(oracle, stage, tmp_path):
    """cPlp cut behind the IDFT (doLP = 0: the autocorrelation) or behind the LP analysis (doLpToCeps = 0: the LP coefficients) --
    round 6, smilehip_plp_stage_frames: config/plp/PLP_0_D_A.conf with those options changed; the plugin's [plp] level and its HTK
    output equal the plain binary's byte for byte, every frame through the override."""
    from test_oracle_pin_plp import plp_conf_cut
    from opensmile_amd import synth
    exe = os.path.join(oracle.REF_DIR, "SMILExtract")
    plug = os.path.join(PLUGDIR, "plugins", "libsmilehip_plugin.so")
    if not (os.path.exists(exe) and os.path.exists(plug)):
        pytest.skip("oracle/_ref/SMILExtract or the plugin .so not built")
    td = str(tmp_path)
    wav = os.path.join(td, "in.wav")
    oracle.write_wav(wav, synth.utterance(5, 48000), 16000)
    c = plp_conf_cut(oracle, stage, td)
    outs = {}
    for mode in ("plain", "plugin"):
        env = dict(os.environ)
        env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "opensmile_amd"), oracle.REF_DIR, env.get("LD_LIBRARY_PATH", "")])
        trace = os.path.join(td, "trace.txt")
        env["SMILEHIP_PLUGIN_TRACE"] = trace
        out = os.path.join(td, f"o_{mode}.htk")
        r = subprocess.run([exe, "-C", c, "-I", wav, "-O", out, "-l", "1"], cwd=PLUGDIR if mode == "plugin" else td, env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.stderr + r.stdout)[-1500:]
        outs[mode] = (open(os.path.join(td, "tap_plp.htk"), "rb").read(), open(out, "rb").read())
        if mode == "plugin":
            tr = dict(l.split() for l in open(trace) if len(l.split()) == 2)
            assert int(tr["cPlp"]) > 0 and int(tr.get("cPlp.cpu", 0)) == 0
    assert len(outs["plain"][0]) > 1000 and outs["plain"] == outs["plugin"]
    got = oracle.read_htk(os.path.join(td, "tap_plp.htk"))[0]
    ref = oracle.plp_static_stage(synth.utterance(5, 48000), stage)
    assert got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("case", ["htk_c0", "htk_first1", "htk_nolog"])
def test_plugin_mfcc_inverse(oracle, case, tmp_path):
    """cMfcc with inverse = 1 (round 6, smilehip_mfcc_inverse_frames): a second cMfcc instance turns the cepstral level of an MFCC file
    back into 26 mel bands; the plugin's level equals the plain binary's byte for byte and the oracle's bit for bit."""
    from test_oracle_pin_mfcc_inverse import CASES, inverse_conf
    from opensmile_amd import synth
    exe = os.path.join(oracle.REF_DIR, "SMILExtract")
    plug = os.path.join(PLUGDIR, "plugins", "libsmilehip_plugin.so")
    if not (os.path.exists(exe) and os.path.exists(plug)):
        pytest.skip("oracle/_ref/SMILExtract or the plugin .so not built")
    td = str(tmp_path)
    wav = os.path.join(td, "in.wav")
    oracle.write_wav(wav, synth.utterance(5, 48000), 16000)
    c = inverse_conf(case, td)
    outs = {}
    for mode in ("plain", "plugin"):
        env = dict(os.environ)
        env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "opensmile_amd"), oracle.REF_DIR, env.get("LD_LIBRARY_PATH", "")])
        trace = os.path.join(td, "trace.txt")
        env["SMILEHIP_PLUGIN_TRACE"] = trace
        r = subprocess.run([exe, "-C", c, "-I", wav, "-O", os.path.join(td, "o.htk"), "-l", "1"], cwd=PLUGDIR if mode == "plugin" else td, env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.stderr + r.stdout)[-1500:]
        outs[mode] = open(os.path.join(td, "tap_m.htk"), "rb").read()
        if mode == "plugin":
            tr = dict(l.split() for l in open(trace) if len(l.split()) == 2)
            assert int(tr["cMfcc"]) > 0 and int(tr.get("cMfcc.cpu", 0)) == 0
    assert len(outs["plain"]) > 1000 and outs["plain"] == outs["plugin"]
    _f, _level, first, last, lifter, htk, dolog = CASES[case]
    cep = oracle.read_htk(os.path.join(td, "tap_c.htk"))[0]
    got = oracle.read_htk(os.path.join(td, "tap_m.htk"))[0]
    ref = oracle.mfcc_inverse_rows(cep, first, last, 26, lifter, htk, dolog)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("case", ["power_htk_257", "mag_257", "power_nohtk_129", "power_htk_band"])
def test_plugin_melspec_inverse(oracle, case, tmp_path):
    """cMelspec with inverse = 1 (round 6, smilehip_melspec_inverse_table_frames): a second cMelspec instance turns the mel level of
    MFCC12_0_D_A.conf back into a magnitude spectrum; the plugin's level equals the plain binary's byte for byte and the oracle's bit for bit."""
    from test_oracle_pin_melspec_inverse import CASES, inverse_conf
    from opensmile_amd import synth
    exe = os.path.join(oracle.REF_DIR, "SMILExtract")
    plug = os.path.join(PLUGDIR, "plugins", "libsmilehip_plugin.so")
    if not (os.path.exists(exe) and os.path.exists(plug)):
        pytest.skip("oracle/_ref/SMILExtract or the plugin .so not built")
    td = str(tmp_path)
    wav = os.path.join(td, "in.wav")
    oracle.write_wav(wav, synth.utterance(5, 48000), 16000)
    c = inverse_conf(case, td)
    outs = {}
    for mode in ("plain", "plugin"):
        env = dict(os.environ)
        env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "opensmile_amd"), oracle.REF_DIR, env.get("LD_LIBRARY_PATH", "")])
        trace = os.path.join(td, "trace.txt")
        env["SMILEHIP_PLUGIN_TRACE"] = trace
        r = subprocess.run([exe, "-C", c, "-I", wav, "-O", os.path.join(td, "o.htk"), "-l", "1"], cwd=PLUGDIR if mode == "plugin" else td, env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.stderr + r.stdout)[-1500:]
        outs[mode] = open(os.path.join(td, "tap_s.htk"), "rb").read()
        if mode == "plugin":
            tr = dict(l.split() for l in open(trace) if len(l.split()) == 2)
            assert int(tr["cMelspec"]) > 0 and int(tr.get("cMelspec.cpu", 0)) == 0
    assert len(outs["plain"]) > 1000 and outs["plain"] == outs["plugin"]
    n_out, power, htk, lo, hi = CASES[case]
    mel = oracle.read_htk(os.path.join(td, "tap_m.htk"))[0]
    got = oracle.read_htk(os.path.join(td, "tap_s.htk"))[0]
    ref = oracle.melspec_inverse_rows(mel, n_out, 512 / 16000.0, lo, hi, power, htk)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
