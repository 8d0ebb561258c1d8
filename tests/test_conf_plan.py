"""smilextract_hip -C file.conf: plans derived from the reference's configuration files by the host library's own reader
(opensmile_amd/host/conf_plan.cpp). CPU part: the eight files of config/mfcc and config/plp map to exactly the presets
(`smilehip_config_htk_variant`), the four big sets are recognised by their graph fingerprint, option changes land in the
right fields, inexpressible graphs / options are refused by name. GPU part: -C runs end to end and equals --set."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "opensmile_amd", "smilextract_hip")
CONF = os.path.join(ROOT, "oracle", "_ref", "config")
G = os.path.join(ROOT, "tests", "golden", "files")

def _copytree(src, dst):
    """a writable copy (the reference tree, and with it oracle/_ref/config, may be read-only; the tests edit their copies)"""
    shutil.copytree(src, dst)
    for d, _dirs, files in os.walk(dst):
        os.chmod(d, 0o755)
        for f in files:
            os.chmod(os.path.join(d, f), 0o644)


needs_conf = pytest.mark.skipif(not (os.path.isdir(CONF) and os.path.exists(EXE)), reason="oracle/_ref/config or smilextract_hip not built")


def describe(conf, *extra):
    r = subprocess.run([EXE, "-C", conf, "--describe", *extra], capture_output=True, text=True)
    kv = dict(l.split("=", 1) for l in r.stdout.split("\n") if "=" in l)
    return r.returncode, kv, r.stderr


@needs_conf
def test_cepstral_confs_map_to_the_presets():
    from opensmile_amd import capi
    for d, name in (("mfcc", "MFCC12_0_D_A"), ("mfcc", "MFCC12_E_D_A"), ("mfcc", "MFCC12_0_D_A_Z"), ("mfcc", "MFCC12_E_D_A_Z"),
                    ("plp", "PLP_0_D_A"), ("plp", "PLP_E_D_A"), ("plp", "PLP_0_D_A_Z"), ("plp", "PLP_E_D_A_Z")):
        rc, kv, err = describe(os.path.join(CONF, d, name + ".conf"))
        assert rc == 0 and kv["preset"] == "", err
        ref = capi.htk_variant_config(name)
        for f in ("chain_kind", "preemph", "preemph_de", "win_func", "zero_pad_symmetric", "n_bands", "use_power", "mel_htk_compatible",
                  "first_mfcc", "mfcc_htk_compatible", "n_delta", "delta_win", "append_log_energy", "cms"):
            assert int(kv[f]) == int(getattr(ref, f)), (name, f)
        for f in ("frame_size_sec", "frame_step_sec", "preemph_k", "win_gain", "win_offset", "lofreq", "hifreq", "cep_lifter"):
            assert float(kv[f]) == pytest.approx(float(getattr(ref, f)), rel=1e-7, abs=0), (name, f)
        if name.startswith("PLP"):
            assert int(kv["plp_lp_order"]) == ref.plp_lp_order and float(kv["plp_compression"]) == pytest.approx(ref.plp_compression, rel=1e-7)
        else:
            assert int(kv["last_mfcc"]) == ref.last_mfcc
        names = kv["names"].split(";")
        n_static = len(names) // 3
        assert names[n_static].replace("_de", "") == names[0] and names[2 * n_static].replace("_de_de", "") == names[0]
        if "_E_" in name:
            assert names[n_static - 1] == "pcm_LOGenergy" and names[-1] == "pcm_LOGenergy_de_de"
    assert {int(describe(os.path.join(CONF, d, n + ".conf"))[1]["parm_kind"]) for d, n in
            (("mfcc", "MFCC12_0_D_A_Z"), ("mfcc", "MFCC12_E_D_A_Z"), ("plp", "PLP_0_D_A_Z"), ("plp", "PLP_E_D_A_Z"))} == {11014, 2886, 11019, 8971}


@needs_conf
def test_big_sets_are_recognised_by_their_graph():
    for rel, preset in (("is09-13/IS09_emotion.conf", "is09_emotion"), ("compare16/ComParE_2016.conf", "compare16"),
                        ("is09-13/IS13_ComParE.conf", "is13_compare"), ("egemaps/v02/eGeMAPSv02.conf", "egemapsv02")):
        rc, kv, err = describe(os.path.join(CONF, rel))
        assert rc == 0 and kv["preset"] == preset, err
    # output options do not change the graph; a changed processing option does
    rc, kv, err = describe(os.path.join(CONF, "is09-13/IS09_emotion.conf"), "-O", "x.arff", "-instname", "a")
    assert rc == 0 and kv["preset"] == "is09_emotion"
    # the two sub-graphs of eGeMAPSv02.conf are presets of their own (column subsets of its levels)
    for rel, preset in (("gemaps/v01b/GeMAPSv01b.conf", "gemapsv01b"), ("egemaps/v01b/eGeMAPSv01b.conf", "egemapsv01b")):
        rc, kv, err = describe(os.path.join(CONF, rel))
        assert rc == 0 and kv["preset"] == preset, err
    # GeMAPSv01a / eGeMAPSv01a have other option values (zeroPadSymmetric, useBrokenJitterThresh, formant maxF): presets of their own
    for rel, preset in (("gemaps/v01a/GeMAPSv01a.conf", "gemapsv01a"), ("egemaps/v01a/eGeMAPSv01a.conf", "egemapsv01a")):
        rc, kv, err = describe(os.path.join(CONF, rel))
        assert rc == 0 and kv["preset"] == preset, err


@needs_conf
def test_option_changes_and_refusals(tmp_path):
    src = os.path.join(CONF, "mfcc", "MFCC12_0_D_A.conf")
    _copytree(os.path.join(CONF, "shared"), tmp_path / "shared")
    os.makedirs(tmp_path / "mfcc")
    txt = open(src).read()

    def variant(name, *subs):
        t = txt
        for sub in subs:
            a, b = sub[0], sub[1]
            assert a in t, a
            t = t.replace(a, b, sub[2]) if len(sub) > 2 else t.replace(a, b)
        p = tmp_path / "mfcc" / name
        p.write_text(t)
        return str(p)

    rc, kv, err = describe(variant("a.conf", ("frameSize = 0.0250", "frameSize = 0.032"), ("nBands = 26", "nBands = 40"),
                                   ("lastMfcc  = 12", "lastMfcc  = 19"), ("hifreq = 8000", "hifreq = 7600"), ("winFunc = ham", "winFunc = Hann"),
                                   ("k = 0.97", "k = 0.95"), ("cepLifter = 22.0", "cepLifter = 0")))
    assert rc == 0, err
    assert float(kv["frame_size_sec"]) == 0.032 and int(kv["n_bands"]) == 40 and int(kv["last_mfcc"]) == 19 and float(kv["hifreq"]) == 7600
    assert int(kv["win_func"]) == 1 and float(kv["preemph_k"]) == pytest.approx(0.95, rel=1e-6) and float(kv["cep_lifter"]) == 0
    assert len(kv["names"].split(";")) == 3 * 20
    # defaults of the reference when a line is absent: cMelspec lofreq 20, cTransformFFT zeroPadSymmetric 1
    rc, kv, err = describe(variant("b.conf", ("lofreq = 0\n", ""), ("zeroPadSymmetric = 0", "")))
    assert rc == 0 and float(kv["lofreq"]) == 20.0 and int(kv["zero_pad_symmetric"]) == 1, err
    # inexpressible option values and unknown options are refused by name
    rc, kv, err = describe(variant("c.conf", ("specScale = mel", "specScale = bark")))
    assert rc != 0 and "specScale" in err
    rc, kv, err = describe(variant("d.conf", ("htkcompatible = 1\nnBands", "htkcompatible = 1\nshowFbank = 1\nnBands")))
    assert rc != 0 and "showFbank" in err
    rc, kv, err = describe(variant("e.conf", ("[accel:cDeltaRegression]\nreader.dmLevel=ft0de", "[accel:cDeltaRegression]\nreader.dmLevel=nowhere")))
    assert rc != 0 and "nowhere" in err
    # an option the file does not define is refused like SMILExtract refuses it
    rc, kv, err = describe(src, "-nosuchoption", "1")
    assert rc != 0 and "nosuchoption" in err
    # options the file defines but this program does not implement are refused (they used to be parsed and ignored:
    # ADVICE r2): a segment of the file, a second sink, another timestamp / header / append setting
    for extra in (("-start", "1"), ("-end", "2"), ("-appendcsv", "1"), ("-timestampcsv", "0"), ("-headercsv", "0"), ("-arffoutput", "a.arff")):
        rc, kv, err = describe(src, *extra)
        assert rc != 0 and extra[0][1:] in err and "does not implement" in err, (extra, err)
    rc, kv, err = describe(src, "-start", "0", "-end", "-1")            # the defaults themselves are fine
    assert rc == 0, err
    big = os.path.join(CONF, "is09-13", "IS09_emotion.conf")
    for extra in (("-lldarffoutput", "a.arff"), ("-timestampcsv", "0"), ("-appendarff", "0"), ("-relation", "x"), ("-frameTimeAdd", "1")):
        rc, kv, err = describe(big, *extra)
        assert rc != 0 and extra[0][1:] in err, (extra, err)
    # the message says whether the refused value is the file's or the component's default: -timestampcsv 0 sets frameTime in
    # [csvsink] (named by its other spelling in the message); a [csvsink] without its frameIndex line falls back to the default 1
    bigdir = tmp_path / "is09-13"
    _copytree(os.path.join(CONF, "is09-13"), bigdir)
    inc = (tmp_path / "shared" / "standard_data_output.conf.inc")
    t = inc.read_text()
    assert "frameIndex=0\n" in t and "frameTime=\\cm[timestampcsv{1}" in t
    inc.write_text(t.replace("frameTime=\\cm[timestampcsv{1}", "frameTime=0 ;\\cm[timestampcsv{1}", 1))
    rc, kv, err = describe(str(bigdir / "IS09_emotion.conf"), "-csvoutput", "x.csv")
    assert rc != 0 and "timestamp / frameTime = 0" in err and "where the file is silent" not in err, err
    inc.write_text(t.replace("frameIndex=0\n", "", 1))
    rc, kv, err = describe(str(bigdir / "IS09_emotion.conf"), "-csvoutput", "x.csv")
    assert rc != 0 and "number / frameIndex = 1" in err and "where the file is silent" in err, err
    inc.write_text(t)
    # source / sink sections edited in the file itself: a segment, a second sink on a stage level, an unknown sink type
    rc, kv, err = describe(variant("f.conf", ("instance[frame].type=cFramer", "instance[frame].type=cFramer\ninstance[waveIn2].type=cWaveSource"), ("[frame:cFramer]", "[waveIn2:cWaveSource]\nstart = 0.5\n[frame:cFramer]")))
    assert rc != 0 and "start" in err, err
    rc, kv, err = describe(variant("g.conf", ("instance[frame].type=cFramer", "instance[frame].type=cFramer\ninstance[tap].type=cCsvSink"),
                                   ("[frame:cFramer]", "[tap:cCsvSink]\nreader.dmLevel=melspec\nfilename=tap.csv\n[frame:cFramer]")))
    assert rc != 0 and "melspec" in err, err


@pytest.mark.gpu
@needs_conf
def test_conf_front_end_equals_set(tmp_path):
    """-C <the reference's file> produces the same bytes as --set <its preset>; a modified file (32 ms frames, 40 bands, 20
    cepstra, Hann window) equals the REAL binary on that file within the chain's tolerance."""
    from test_host_io import read_htk
    wav = os.path.join(G, "u3_4000.wav")
    for rel, setname, opts in (("mfcc/MFCC12_0_D_A.conf", "mfcc12_0_d_a", ["-O"]), ("plp/PLP_E_D_A_Z.conf", "plp_e_d_a_z", ["-O"]),
                               ("egemaps/v02/eGeMAPSv02.conf", "egemapsv02", ["-htkoutput"])):
        a, b = str(tmp_path / "a.htk"), str(tmp_path / "b.htk")
        subprocess.run([EXE, "-C", os.path.join(CONF, rel), "-I", wav, opts[0], a], check=True)
        subprocess.run([EXE, "--set", setname, "-I", wav, opts[0], b], check=True)
        assert open(a, "rb").read() == open(b, "rb").read(), rel
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "SMILExtract")
    if not os.path.exists(ref_exe):
        pytest.skip("oracle/_ref/SMILExtract not built")
    _copytree(os.path.join(CONF, "shared"), tmp_path / "shared")
    os.makedirs(tmp_path / "mfcc")
    t = open(os.path.join(CONF, "mfcc", "MFCC12_0_D_A.conf")).read()
    for x, y in (("frameSize = 0.0250", "frameSize = 0.032"), ("nBands = 26", "nBands = 40"), ("lastMfcc  = 12", "lastMfcc  = 19"),
                 ("winFunc = ham", "winFunc = Hann"), ("lofreq = 0\n", "lofreq = 100\n")):
        assert x in t
        t = t.replace(x, y)
    conf = str(tmp_path / "mfcc" / "mod.conf")
    open(conf, "w").write(t)
    a, b = str(tmp_path / "m_hip.htk"), str(tmp_path / "m_ref.htk")
    subprocess.run([EXE, "-C", conf, "-I", wav, "-O", a], check=True)
    env = dict(os.environ, SMILEHIP_PLUGIN_COMPONENTS="none")
    subprocess.run([ref_exe, "-C", conf, "-I", wav, "-O", b, "-l", "0"], check=True, env=env, cwd=str(tmp_path))
    ha, xa = read_htk(a)
    hb, xb = read_htk(b)
    assert ha == hb and xa.shape == xb.shape and xa.shape[1] == 60
    scale = np.abs(xb[:, :20]).max(axis=1, keepdims=True)
    assert (np.abs(xa - xb) / scale).max() <= 1e-5


@needs_conf
def test_every_reference_conf_gets_a_plan_or_a_named_refusal():
    """All configuration files of the reference's config/ tree through `smilextract_hip -C <file> --describe`: the fourteen the fused
    path covers produce a plan; every other one is refused with a message that names what is not covered (a component
    instance, an option, a missing section) -- never a crash, a hang or a silent default."""
    confs = sorted(os.path.join(d, f) for d, _, fs in os.walk(CONF) for f in fs if f.endswith(".conf"))
    assert len(confs) >= 60
    planned = []
    for c in confs:
        r = subprocess.run([EXE, "-C", c, "--describe"], capture_output=True, text=True, timeout=60)
        rel = os.path.relpath(c, CONF)
        assert r.returncode in (0, 1, 2), (rel, r.returncode, r.stderr[-300:])          # no signal, no abort
        if r.returncode == 0:
            planned.append(rel)
            assert "chain_kind=" in r.stdout or "preset=" in r.stdout, rel
        else:
            msg = r.stderr.strip()
            assert msg.startswith("smilextract_hip:") and len(msg) > 40, (rel, msg)
            assert "cannot run on the fused path" in msg or "outside a section" in msg or "[" in msg, (rel, msg)
    assert set(planned) == {"mfcc/MFCC12_0_D_A.conf", "mfcc/MFCC12_0_D_A_Z.conf", "mfcc/MFCC12_E_D_A.conf", "mfcc/MFCC12_E_D_A_Z.conf",
                            "plp/PLP_0_D_A.conf", "plp/PLP_0_D_A_Z.conf", "plp/PLP_E_D_A.conf", "plp/PLP_E_D_A_Z.conf",
                            "is09-13/IS09_emotion.conf", "is09-13/IS13_ComParE.conf", "compare16/ComParE_2016.conf",
                            "egemaps/v02/eGeMAPSv02.conf", "gemaps/v01b/GeMAPSv01b.conf", "egemaps/v01b/eGeMAPSv01b.conf",
                                "gemaps/v01a/GeMAPSv01a.conf", "egemaps/v01a/eGeMAPSv01a.conf"}


@needs_conf
def test_edited_big_set_files_are_the_set_with_other_parameters(tmp_path):
    """ComParE_2016.conf / eGeMAPSv02.conf with other values of the F0 group's options are recognised through the masked
    fingerprint (the graph is the shipped one, only parameter options differ) and --describe lists the values it will run
    with; an edit of any other option is still refused."""
    _copytree(CONF, tmp_path / "config")
    inc = tmp_path / "config" / "compare16" / "ComParE_2016_core.lld.conf.inc"
    t = inc.read_text()
    for a, b in (("maxPitch = 620", "maxPitch = 500"), ("minPitch = 52", "minPitch = 60"), ("nHarmonics = 15", "nHarmonics = 12"),
                 ("bufferLength=30", "bufferLength=20"), ("searchRangeRel = 0.250000", "searchRangeRel = 0.2"),
                 ("voicingCutoff = 0.700000", "voicingCutoff = 0.65"), ("threshold=0.001", "threshold=0.002")):
        assert a in t, a
        t = t.replace(a, b)
    inc.write_text(t)
    rc, kv, err = describe(str(tmp_path / "config" / "compare16" / "ComParE_2016.conf"))
    assert rc == 0 and kv["preset"] == "compare16", err
    assert float(kv["param.pitch_max"]) == 500 and float(kv["param.pitch_min"]) == 60 and float(kv["param.shs_n_harmonics"]) == 12
    assert float(kv["param.vit_buffer_len"]) == 20 and float(kv["param.jitter_search_range"]) == 0.2
    assert float(kv["param.voicing_cutoff"]) == 0.65 and float(kv["param.f0_min_energy"]) == pytest.approx(0.002)
    inc.write_text(t.replace("nCandidates = 6", "nCandidates = 5"))            # not a parameter of the kernels: refused
    rc, kv, err = describe(str(tmp_path / "config" / "compare16" / "ComParE_2016.conf"))
    assert rc != 0


@pytest.mark.gpu
@needs_conf
def test_edited_big_set_files_equal_the_binary_on_the_same_file(tmp_path):
    """The edited files through smilextract_hip -C against the REAL binary on the same edited files: LLD level and
    functionals bit for bit."""
    from test_host_io import read_htk
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "SMILExtract")
    if not os.path.exists(ref_exe):
        pytest.skip("oracle/_ref/SMILExtract not built")
    import wave
    from opensmile_amd import synth
    wav = str(tmp_path / "u.wav")
    with wave.open(wav, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(synth.utterance(7, 64000).tobytes())
    _copytree(CONF, tmp_path / "config")
    edits = {
        "compare16/ComParE_2016_core.lld.conf.inc": (("maxPitch = 620", "maxPitch = 500"), ("minPitch = 52", "minPitch = 60"),
                                                    ("nHarmonics = 15", "nHarmonics = 12"), ("bufferLength=30", "bufferLength=20"),
                                                    ("searchRangeRel = 0.250000", "searchRangeRel = 0.2")),
        "gemaps/v01b/GeMAPSv01b_core.lld.conf.inc": (("maxPitch = 1000", "maxPitch = 800"), ("nHarmonics = 15", "nHarmonics = 10")),
    }
    for rel, subs in edits.items():
        p = tmp_path / "config" / rel
        t = p.read_text()
        for a, b in subs:
            assert a in t, (rel, a)
            t = t.replace(a, b)
        p.write_text(t)
    for rel in ("compare16/ComParE_2016.conf", "egemaps/v02/eGeMAPSv02.conf"):
        conf = str(tmp_path / "config" / rel)
        outs = {}
        for tag, exe, extra, env in (("hip", EXE, [], None), ("ref", ref_exe, ["-l", "0"], dict(os.environ, SMILEHIP_PLUGIN_COMPONENTS="none"))):
            lld, fun = str(tmp_path / f"{tag}.lld.htk"), str(tmp_path / f"{tag}.func.htk")
            subprocess.run([exe, "-C", conf, "-I", wav, "-lldhtkoutput", lld, "-htkoutput", fun] + extra, check=True, env=env, cwd=str(tmp_path))
            outs[tag] = (read_htk(lld), read_htk(fun))
        for k in (0, 1):
            (ha, xa), (hb, xb) = outs["hip"][k], outs["ref"][k]
            assert ha == hb and xa.shape == xb.shape, rel
            assert np.array_equal(xa.view(np.uint32), xb.view(np.uint32)), (rel, k, np.abs(xa - xb).max())


def _edit_compare16(tmp_path, is13=False):
    """ComParE_2016.conf (or IS13_ComParE.conf) with maxPitch changed, lastMfcc = 12, Moments removed from is13_functionalsB,
    Peaks2 removed from is13_functionalsLLD, Regression removed from is13_functionalsNz."""
    _copytree(CONF, tmp_path / "config")
    d = "is09-13" if is13 else "compare16"
    stem = "IS13_ComParE" if is13 else "ComParE_2016"
    p = tmp_path / "config" / d / (stem + "_core.lld.conf.inc")
    t = p.read_text()
    for a, b in (("maxPitch = 620", "maxPitch = 550"), ("lastMfcc  = 14", "lastMfcc  = 12")):
        assert a in t, a
        t = t.replace(a, b)
    p.write_text(t)
    p = tmp_path / "config" / d / (stem + "_core.func.conf.inc")
    t = p.read_text()
    subs = (("functionalsEnabled = Extremes ; Percentiles ; Moments ; Segments ; Times ;  Lpc", "functionalsEnabled = Extremes ; Percentiles ; Segments ; Times ;  Lpc", 1),
            ("functionalsEnabled = Means ; Peaks2 ; Regression", "functionalsEnabled = Means ; Regression", 0),
            ("functionalsEnabled = Means ; Extremes ; Regression ; Percentiles ; Moments ; Times ; Lpc", "functionalsEnabled = Means ; Extremes ; Percentiles ; Moments ; Times ; Lpc", 0))
    for a, b, which in subs:
        assert a in t, a
        i = -1
        for _ in range(which + 1):
            i = t.index(a, i + 1)
        t = t[:i] + b + t[i + len(a):]                  # the which-th occurrence (A and B share a line: B is the second)
    p.write_text(t)
    return str(tmp_path / "config" / d / (stem + ".conf"))


@needs_conf
def test_output_selection_edits_are_recognised(tmp_path):
    conf = _edit_compare16(tmp_path)
    rc, kv, err = describe(conf)
    assert rc == 0 and kv["preset"] == "compare16", err
    # a family the shipped instance does not have: refused by name
    p = tmp_path / "config" / "compare16" / "ComParE_2016_core.func.conf.inc"
    p.write_text(p.read_text().replace("functionalsEnabled = Means ; Regression", "functionalsEnabled = Means ; Regression ; Onset"))
    r = subprocess.run([EXE, "-C", conf, "-I", "x.wav", "-lldhtkoutput", "x.htk"], capture_output=True, text=True)
    assert r.returncode != 0 and "Onset" in r.stderr and "functionalsEnabled" in r.stderr, r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("is13", [False, True])
def test_output_selection_edits_equal_the_binary(tmp_path, is13):
    """lastMfcc = 12 and functional families left out (with an F0 parameter changed as well): smilextract_hip -C on the edited file
    against the REAL binary on the same file -- names, LLD level and functionals bit for bit."""
    from test_host_io import read_htk
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "SMILExtract")
    if not os.path.exists(ref_exe):
        pytest.skip("oracle/_ref/SMILExtract not built")
    import wave
    from opensmile_amd import synth
    conf = _edit_compare16(tmp_path, is13)
    wav = str(tmp_path / "u.wav")
    with wave.open(wav, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(synth.utterance(9, 50000).tobytes())
    outs = {}
    for tag, exe, extra, env in (("hip", EXE, [], None), ("ref", ref_exe, ["-l", "0"], dict(os.environ, SMILEHIP_PLUGIN_COMPONENTS="none"))):
        lld, fun, csv, fcsv = (str(tmp_path / f"{tag}{e}") for e in (".lld.htk", ".func.htk", ".lld.csv", ".func.csv"))
        subprocess.run([exe, "-C", conf, "-I", wav, "-lldhtkoutput", lld, "-htkoutput", fun, "-lldcsvoutput", csv, "-csvoutput", fcsv,
                        "-instname", "u"] + extra, check=True, env=env, cwd=str(tmp_path))
        outs[tag] = (read_htk(lld), read_htk(fun), open(csv).readline(), open(fcsv).readline())
    assert outs["hip"][2] == outs["ref"][2] and outs["hip"][3] == outs["ref"][3]         # element names
    for k in (0, 1):
        (ha, xa), (hb, xb) = outs["hip"][k], outs["ref"][k]
        assert ha == hb and xa.shape == xb.shape
        assert np.array_equal(xa.view(np.uint32), xb.view(np.uint32)), (k, np.abs(xa - xb).max())
    assert outs["ref"][0][1].shape[1] == 126 and outs["ref"][1][1].shape[1] < 6373
