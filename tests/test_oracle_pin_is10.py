"""Pin oracle/lld_oracle_is10.c (cIntensity, cLsp, cPitchSmoother, cVectorOperation) and the Onset family of
oracle/lld_oracle_funcspec.c against the REAL reference binary: the unmodified config/is09-13/IS10_paraling.conf with HTK taps
on its internal levels (oracle/conf/is10_taps.conf); every restated component applied to the binary's own input level must
reproduce the binary's output level bit for bit."""
import numpy as np
import pytest

from oracle import lldo

pytestmark = pytest.mark.skipif(not lldo.have_ref(), reason="oracle/_ref not built")

CASES = [(2, 16000), (10, 32000), (5, 9000), (3, 24000), (12, 40000), (7, 2000)]


def same(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))))


@pytest.fixture(scope="module")
def runs():
    from opensmile_amd import synth
    return [(u, n, lldo.run_reference_is10(synth.utterance(u, n))) for u, n in CASES]


def spec_l1(nz):
    """[is10_functL1] / [is10_functL1nz] of IS10_paraling_core.func.conf.inc"""
    s = lldo.FuncSpec()
    lldo._spec_common(s, ["Extremes", "Regression", "Moments", "Percentiles", "Times"])
    s.ext_mask = lldo._mask(lldo.EXT_NAMES, ["maxPos", "minPos", "amean"])
    s.ext_norm = lldo.NORM["frame"]
    s.reg_mask = lldo._mask(lldo.REG_NAMES, ["linregc1", "linregc2", "linregerrA", "linregerrQ"])
    s.mom_mask = lldo._mask(lldo.MOM_NAMES, ["stddev", "skewness", "kurtosis"])
    s.pct_mask = 0x3f
    s.pct_interp = 1
    if nz:
        s.n_pctl, s.n_range = 1, 0
        s.pctl[0] = 0.99
        s.non_zero_functs = 1
    else:
        s.n_pctl, s.n_range = 2, 1
        s.pctl[0], s.pctl[1] = 0.01, 0.99
        s.range_a[0], s.range_b[0] = 0, 1
    s.times_mask = lldo._mask(lldo.TIMES_NAMES, ["upleveltime75", "upleveltime90"])
    s.times_norm = lldo.NORM["segment"]
    return s


def spec_onsets():
    s = lldo.FuncSpec()
    lldo._spec_common(s, ["Onset", "Times"])
    s.ons_mask = 1 << 4
    s.ons_norm = lldo.NORM["segment"]
    s.times_mask = 1 << 12
    s.times_norm = lldo.NORM["second"]
    return s


def test_components_bit_exact(runs):
    for u, n, r in runs:
        what = f"u{u}_{n}"
        assert same(lldo.intensity_rows(r["is10_frames"]), r["is10_intens"]), what
        assert same(lldo.specresample_rows(r["is10_fftc"], 512 / 16000.0, 400 / 16000.0, 1 / 16000.0, 11000.0), r["is10_outpR"]), what
        assert same(lldo.egemaps_lpc_rows(r["is10_outpR"], 8), r["is10_lpc"]), what
        assert same(lldo.lsp_rows(r["is10_lpc"]), r["is10_lsp"]), what
        assert same(lldo.vecop_rows(r["is10_mspec2"], "log"), r["is10_mspec2log"]), what
        shs = r["is10_pitchShs"]                       # nCandidates | F0Cand[6] | candVoicing[6] | candScores[6]
        if shs.size:
            assert shs.shape[1] == 19
            assert same(lldo.pitch_smoother_rows(shs[:, 1:19], flags=2 | 8), r["is10_pitch"].reshape(-1, 2)), what
            assert same(lldo.pitch_smoother_rows(shs[:, 1:19], flags=1), r["is10_pitchF"].reshape(-1, 1)), what


def test_functionals_bit_exact_and_the_rows_they_read(runs):
    """The three cFunctionals instances: which rows of their input levels they summarise at the end of the input is the
    component manager's tick order (measured here: T - 3 of the T rows the smoothed level holds; the onsets instance every row)."""
    for u, n, r in runs:
        what = f"u{u}_{n}"
        T = r["is10_lld1"].shape[0]
        if T < 4:
            continue
        x1 = np.concatenate([r["is10_lld1"][:T - 3], r["is10_lld1_de"][:T - 3]], axis=1)
        assert same(lldo.funcspec(x1, spec_l1(False)).reshape(1, -1), r["is10_funct"]), what
        x2 = np.concatenate([r["is10_lld2"][:T - 3], r["is10_lld2_de"][:T - 3]], axis=1)
        assert same(lldo.funcspec(x2, spec_l1(True)).reshape(1, -1), r["is10_functNz"]), what
        assert same(lldo.funcspec(r["is10_pitchF"], spec_onsets()).reshape(1, -1), r["is10_functOnsets"]), what
        assert same(np.concatenate([r["is10_funct"], r["is10_functNz"], r["is10_functOnsets"]], axis=1), r["func"]), what


def test_onset_family_options():
    """thresholds, useAbsVal, every output and norm against a direct restatement of functionalOnset.cpp:83-151"""
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((300, 7)) * (rng.random((300, 7)) > 0.4)).astype(np.float32)
    for norm in ("segment", "second", "frame"):
        for use_abs, th_on, th_off in ((0, 0.0, 0.0), (1, 0.3, 0.3), (0, 0.5, -0.2), (1, 0.1, 0.6)):
            s = lldo.FuncSpec()
            lldo._spec_common(s, ["Onset"])
            s.ons_mask, s.ons_norm, s.ons_use_abs = 0x1f, lldo.NORM[norm], use_abs
            s.ons_thr_on, s.ons_thr_off = th_on, th_off
            got = lldo.funcspec(x, s)
            for c in range(x.shape[1]):
                col, N = x[:, c], x.shape[0]
                on_pos = off_pos = -1
                n_on = n_off = 0
                oo = 1 if col[0] > np.float32(th_on) else 0
                for i in range(1, N):
                    cur = abs(col[i]) if use_abs else col[i]
                    if cur > np.float32(th_on) and oo == 0:
                        n_on += 1
                        on_pos = i if on_pos == -1 else on_pos
                        oo = 1
                    if cur <= np.float32(th_off) and oo == 1:
                        n_off += 1
                        off_pos = i
                        oo = 0
                off_pos = N - 1 if off_pos == -1 else off_pos
                on_pos = 0 if on_pos == -1 else on_pos
                f32 = np.float32
                T = f32(0.01)
                if norm == "segment":
                    pos = [f32(on_pos) / f32(N), f32(off_pos) / f32(N)]
                elif norm == "second":
                    pos = [f32(on_pos) * T, f32(off_pos) * T]
                else:
                    pos = [f32(on_pos), f32(off_pos)]
                ref = np.array(pos + [f32(n_on), f32(n_off), f32(n_on) / (f32(N) * T)], np.float32)
                assert same(got[c], ref), (norm, use_abs, th_on, th_off, c)


def test_peaks_family_bit_exact(tmp_path):
    """The older peak picker ("Peaks": IS11_speaker_state's family) of oracle/lld_oracle_funcspec.c against the binary on energy
    contours (tests/conf/peaks_family.conf: all five values, the three time norms)."""
    import os
    import subprocess
    from opensmile_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for u, n in ((9, 48000), (3, 16000), (12, 8000), (6, 1200)):
        wav = str(tmp_path / "p.wav")
        lldo.write_wav(wav, synth.utterance(u, n), 16000)
        subprocess.run([os.path.join(lldo.REF_DIR, "SMILExtract"), "-C", os.path.join(root, "tests", "conf", "peaks_family.conf"), "-I", wav,
                        "-T", str(tmp_path), "-l", "0"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        x = lldo.read_htk(str(tmp_path / "tap_energy.htk"))[0]
        for tag, mask, norm in (("seg", 0x1f, "segment"), ("sec", 0x1e, "second"), ("fra", 0x13, "frame")):
            ref = lldo.read_htk(str(tmp_path / ("tap_%s.htk" % tag)))[0]
            s = lldo.FuncSpec()
            lldo._spec_common(s, ["Peaks"])
            s.pko_mask, s.pko_norm = mask, lldo.NORM[norm]
            assert same(lldo.funcspec(x, s).reshape(1, -1), ref), (u, n, tag)


def test_crossings_dct_samples_families_bit_exact(tmp_path):
    """The families Crossings, DCT and Samples of oracle/lld_oracle_funcspec.c against the binary (tests/conf/families_misc.conf)."""
    import os
    import subprocess
    from opensmile_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def spec(fam, **kw):
        s = lldo.FuncSpec()
        lldo._spec_common(s, [fam])
        for k, v in kw.items():
            if k == "sample_pos":
                for i, p in enumerate(v):
                    s.sample_pos[i] = p
            else:
                setattr(s, k, v)
        return s
    cases = {"f_crs": spec("Crossings", crs_mask=7), "f_crs2": spec("Crossings", crs_mask=2), "f_dct": spec("DCT", dct_first=1, dct_last=6),
             "f_dct2": spec("DCT", dct_first=0, dct_last=3), "f_smp": spec("Samples", n_samples=5, sample_pos=[0, .25, .5, .75, 1.0]),
             "f_smp2": spec("Samples", n_samples=5, sample_pos=[0.0, 0.33, 0.5, 0.999, 1.0])}
    for u, n in ((9, 48000), (3, 16000), (6, 1200)):
        wav = str(tmp_path / "p.wav")
        lldo.write_wav(wav, synth.utterance(u, n), 16000)
        subprocess.run([os.path.join(lldo.REF_DIR, "SMILExtract"), "-C", os.path.join(root, "tests", "conf", "families_misc.conf"), "-I", wav,
                        "-T", str(tmp_path), "-l", "0"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        x = lldo.read_htk(str(tmp_path / "tap_energy.htk"))[0]
        for k, s in cases.items():
            ref = lldo.read_htk(str(tmp_path / ("tap_%s.htk" % k)))[0]
            assert same(lldo.funcspec(x, s).reshape(1, -1), ref), (u, n, k)


SEG_CASES = {   # instance of tests/conf/segments_family.conf -> (algorithm, norm, keyword options)
    "delta": ("DELTA", "segment", {}), "deltar": ("DELTA", "frame", dict(ravg=5, rrt=0.1, min_lng=4)),
    "delt2": ("DELTA2", "second", dict(max_num=50, rrt=0.15)), "rel3": ("RELTH", "frame", dict(th=(0.1, 0.5, 0.9), max_num=40)),
    "mrel": ("MRELTH", "second", dict(th=(0.5, 1.5), max_num=60)), "abs": ("ABSTH", "frame", dict(th=(0.05,))),
    "narel": ("NARELTH", "second", dict(th=(0.25,), max_num=100)),
    "narel2": ("NARELTH", "segment", dict(th=(0.0, 0.4, 1.0), min_lng=1, max_num=8)),         # -0.5 and 1.7 clamped (:190-199)
    "namrel": ("NAMRELTH", "frame", dict(th=(1.0,), max_num=30)), "namrel2": ("NAMRELTH", "segment", dict(th=(0.8, 1.2))),
    "naabs": ("NAABSTH", "frame", dict(th=(0.01, 0.05, -3.0), max_num=100)), "chx": ("CHX", "frame", dict(max_num=100)),
    "chx2": ("CHX", "second", dict(xrel=1, min_lng=2, pause=1, max_num=6)),
    "ltx": ("DELTA", "frame", {}), "geqx": ("DELTA", "second", dict(max_num=10, rrt=0.05)),   # no code of their own: delta (:872-874)
}
SEG_ALGO = dict(RELTH=0, NONX=1, EQX=2, MRELTH=3, ABSTH=4, NARELTH=5, NAMRELTH=6, NAABSTH=7, DELTA=8, DELTA2=9, CHX=10)


def segments_spec(algo, norm, th=(), max_num=20, min_lng=None, pause=2, x=0.0, xrel=0, ravg=0, rrt=0.2):
    s = lldo.FuncSpec()
    lldo._spec_common(s, ["Segments"])
    s.period = 0.01
    s.seg_mask, s.seg_norm, s.seg_algo, s.seg_max_num = 0x1f, lldo.NORM[norm], SEG_ALGO[algo], max_num
    s.seg_min_lng, s.seg_auto_min_lng = (3, 1) if min_lng is None else (min_lng, 0)
    s.seg_pause_min_lng, s.seg_x, s.seg_x_is_rel, s.seg_n_thresholds = pause, x, xrel, len(th)
    for i, v in enumerate(th):
        s.seg_thresholds[i] = v
    s.seg_ravg_lng, s.seg_range_rel_threshold = ravg, rrt
    return s


def segments_pcm(u, n):
    """An utterance with silent stretches: RMS energy exactly 0 and log energy at its floor there (chX needs equal values)."""
    from opensmile_amd import synth
    x = synth.utterance(u, n).copy()
    for a, b in ((n // 6, n // 4), (n // 2 - 300, n // 2 + 200), (5 * n // 8, 3 * n // 4)):
        x[a:b] = 0
    return x


def test_segments_every_algorithm_bit_exact(tmp_path):
    """Every segmentationAlgorithm of cFunctionalSegments (functionalSegments.cpp:118-155: delta, delt2, (m)(NA)relTh, (NA)absTh,
    chX, and the names that fall back to delta) against the binary, on RMS and log energy (tests/conf/segments_family.conf)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    conf = os.path.join(root, "tests", "conf", "segments_family.conf")
    found = 0
    for u, n in ((9, 48000), (3, 16000), (6, 1200), (71, 100000)):
        R = _run_taps(conf, segments_pcm(u, n), tmp_path, False)
        x = R("energy")
        for k, (algo, norm, kw) in SEG_CASES.items():
            ref = R("s_" + k)
            assert same(lldo.funcspec(x, segments_spec(algo, norm, **kw)).reshape(1, -1), ref), (u, n, k)
            found += int(ref[0, 0] > 0)
    assert found > 40


def quot_time_cases():
    """instance of tests/conf/quotients_leveltimes.conf -> spec"""
    def pct(mask, interp, pctl, ranges, quots):
        s = lldo.FuncSpec()
        lldo._spec_common(s, ["Percentiles"])
        s.pct_mask, s.pct_interp, s.n_pctl, s.n_range, s.n_quot = mask, interp, len(pctl), len(ranges), len(quots)
        for i, v in enumerate(pctl):
            s.pctl[i] = v
        for i, (a, b) in enumerate(ranges):
            s.range_a[i], s.range_b[i] = a, b
        for i, (a, b) in enumerate(quots):
            s.quot_a[i], s.quot_b[i] = a, b
        return s

    def tm(mask, norm, ul, dl):
        s = lldo.FuncSpec()
        lldo._spec_common(s, ["Times"])
        s.period = 0.01
        s.times_mask, s.times_norm, s.n_ul, s.n_dl = mask, lldo.NORM[norm], len(ul), len(dl)
        s.times_buggy_sec_norm = 1                          # the option's default (functionalTimes.cpp:76)
        for i, v in enumerate(ul):
            s.ul[i] = min(1.0, max(0.0, v))
        for i, v in enumerate(dl):
            s.dl[i] = min(1.0, max(0.0, v))
        return s
    return {"pq": pct(0x07, 1, (0.05, 0.5, 0.95), ((0, 2),), ((0, 2), (2, 0), (1, 1))),
            "pq_norange": pct(0x38, 1, (0.1, 0.9), (), ((1, 0),)),
            "pq_nointerp": pct(0, 0, (0.0, 1.0, 0.3), ((0, 1), (2, 1)), ((1, 0), (0, 1))),
            "tm": tm(0x1fff, "segment", (0.1, 0.6, 1.5), (0.33, -0.2)),
            "tm_sec": tm(0, "second", (0.5,), (0.5, 0.05)),
            "tm_frame": tm(0x1fff, "frame", (0.25, 0.75), ())}


def test_percentile_quotients_and_level_times_bit_exact(tmp_path):
    """Percentiles.pctlquotient[] (functionalPercentiles.cpp:402-411: only under the range switch, zero numerators give 0, the soft
    limiter at 50 / 100) and Times.upleveltime[] / downleveltime[] (functionalTimes.cpp:347-364) against the binary, on RMS and log
    energy with silent stretches (tests/conf/quotients_leveltimes.conf)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    conf = os.path.join(root, "tests", "conf", "quotients_leveltimes.conf")
    cases = quot_time_cases()
    for u, n in ((9, 48000), (3, 16000), (6, 1200), (71, 100000)):
        R = _run_taps(conf, segments_pcm(u, n), tmp_path, False)
        x = R("energy")
        for k, s in cases.items():
            ref = R("f_" + k)
            got = lldo.funcspec(x, s).reshape(1, -1)
            assert got.shape == ref.shape, (k, got.shape, ref.shape)
            assert same(got, ref), (u, n, k, got, ref)


def modulation_cases():
    """instance of tests/conf/modulation_family.conf -> lldo.ModSpecCfg"""
    return {"ms": lldo.modspec_config(),
            "ms_fr": lldo.modspec_config(win_frames=256, step_frames=100, num_bins=20, min_freq=1.0, max_freq=16.0, win_func=1),
            "ms_nz": lldo.modspec_config(win_sec=2.0, step_sec=1.0, resolution=1.0, remove_nz_mean=1, win_func=0)}


def test_modulation_spectrum_bit_exact(tmp_path):
    """cFunctionalModulation (the Modulation family's ModulationSpec values: windowed STFT magnitudes of the contour, natural cubic
    spline onto the modulation-frequency axis, average over the windows) against the binary: the default option set, a frame-based
    set with a Hann window and its own axis, removeNonZeroMean with a rectangular window; contours of 98 .. 998 frames (one short
    window zero-padded to its own power of two; several windows with a dropped short tail)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    conf = os.path.join(root, "tests", "conf", "modulation_family.conf")
    for u, n in ((9, 48000), (71, 160000), (3, 16000), (6, 70000)):
        R = _run_taps(conf, segments_pcm(u, n), tmp_path, False)
        x = R("energy")
        for k, c in modulation_cases().items():
            ref = R("f_" + k)
            got = lldo.modspec(x, c).reshape(1, -1)
            assert got.shape == ref.shape and same(got, ref), (u, n, k)


def _run_taps(conf, pcm, tmp_path, cwd_taps):
    import os
    import subprocess
    wav = str(tmp_path / "in.wav")
    lldo.write_wav(wav, pcm, 16000)
    cmd = [os.path.join(lldo.REF_DIR, "SMILExtract"), "-C", conf, "-I", wav, "-l", "0"] + ([] if cwd_taps else ["-T", str(tmp_path)])
    subprocess.run(cmd, check=True, cwd=str(tmp_path), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return lambda k: lldo.read_htk(str(tmp_path / ("tap_%s.htk" % k)))[0]


def test_pitch_smoother_option_paths_bit_exact(tmp_path):
    """octaveCorrection = 1 (with simple post smoothing and with none; all four outputs): paths no shipped configuration uses, on
    IS10_paraling's own candidates (oracle/conf/pitch_smoother_options.conf)."""
    import os
    from opensmile_amd import synth
    conf = os.path.join(os.path.dirname(lldo.IS10_TAPS_CONF), "pitch_smoother_options.conf")
    changed = 0
    for u, n in ((9, 48000), (71, 24000), (4, 30000)):
        R = _run_taps(conf, synth.utterance(u, n), tmp_path, True)
        shs = R("shs")
        for tag, simple in (("oct", 1), ("none", 0)):
            got = lldo.pitch_smoother_rows(shs[:, 1:19], 6, 0.7, 1, simple, 15)
            assert same(got, R(tag)), (u, n, tag)
            changed += int((lldo.pitch_smoother_rows(shs[:, 1:19], 6, 0.7, 0, simple, 15) != got).any(axis=1).sum())
    assert changed > 0                                      # the correction did act on some frames


def test_operator_option_paths_bit_exact(tmp_path):
    """Every element-wise and vector-sum operation of cVectorOperation, cIntensity with both outputs, cLsp on 10 and 16 LP
    coefficients (tests/conf/is10_ops_options.conf)."""
    import os
    from opensmile_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    conf = os.path.join(root, "tests", "conf", "is10_ops_options.conf")
    for u, n in ((9, 48000), (3, 16000)):
        R = _run_taps(conf, synth.utterance(u, n), tmp_path, False)
        assert same(lldo.intensity_rows(R("frames"), 1, 1), R("intens"))
        melc = R("melc")
        assert same(lldo.vecop_rows(R("mel"), "lgA", 0.5), melc) and melc.min() < 0 < melc.max()
        for op, p in (("add", 0.37), ("mul", -2.5), ("log", 1), ("lgA", 10), ("sqr", 1), ("ee", 1), ("abs", 1), ("dBp", 1), ("dBv", 1)):
            assert same(lldo.vecop_rows(melc, op, p), R("vo_" + op)), op
        for op in ("sum", "ssm", "ll1", "ll2"):
            assert same(lldo.vecop_reduce_rows(melc, op).reshape(-1, 1), R("vo_" + op)), op
        for p in (10, 16):
            assert same(lldo.lsp_rows(R("lpc%d" % p)), R("lsp%d" % p)), p


def test_jitter_behind_pitch_smoother_bit_exact(runs):
    """cPitchJitter as IS10_paraling configures it (searchRangeRel 0.2, useBrokenJitterThresh's default 1) on F0 frames that come from
    cPitchSmoother: the value of frame t carries the time stamp of frame t + 1 (lldo_set_jitter_time_shift) -- the binary's level."""
    import ctypes as C
    from opensmile_amd import synth
    L = lldo.lib()
    L.lldo_pitch_jitter_ex.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_double, C.c_double, C.c_double,
                                       C.c_void_p, C.c_void_p]
    L.lldo_set_jitter_time_shift.argtypes = [C.c_long]
    for u, n, r in runs:
        f0 = np.ascontiguousarray(r["is10_pitchF"].reshape(-1))
        if not len(f0):
            continue
        x = (synth.utterance(u, n).astype(np.float32) / np.float32(32767.0)).astype(np.float32)
        out = np.zeros((len(f0), 4), np.float32)
        lldo.compare_set_is13(True)
        L.lldo_set_jitter_time_shift(1)
        try:
            L.lldo_pitch_jitter_ex(x.ctypes.data, len(x), f0.ctypes.data, len(f0), 960, 160, 16000.0, 0.010, 0.2, out.ctypes.data, None)
        finally:
            lldo.compare_set_is13(False)
            L.lldo_set_jitter_time_shift(0)
        assert same(out[:, :3], r["is10_jitter"]), (u, n)


@pytest.mark.parametrize("u,n", [(71, 24000), (2, 16000), (10, 32000), (5, 9000), (12, 40000), (7, 2000), (9, 160000)])
def test_whole_is10_chain_from_the_samples_bit_exact(u, n):
    """lldo.is10_lld_chain / is10_func: IS10_paraling.conf restated from the PCM samples on -- both framers, every component, the
    smoothers' and delta instances' end-of-input behaviour over input levels of different lengths, the onlyInSegments delta whose norm
    grows over the file, the rows the functionals read -- equals the binary's LLD file (76 columns) and functionals (1582) bit for
    bit. This is the oracle a fused IS10 chain will be held against. Utterances with fewer than four 60 ms frames
    follow other end-of-input rules and are not covered."""
    from opensmile_amd import synth
    pcm = synth.utterance(u, n)
    r = lldo.run_reference_is10(pcm, levels=[])
    assert same(lldo.is10_lld_chain(pcm), r["lld"])
    assert same(lldo.is10_func(pcm), r["func"])
