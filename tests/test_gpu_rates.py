"""The big sets at sample rates other than 16 kHz (missing 2): every frame size of the shipped configs is given in
seconds, so ComParE_2016 / eGeMAPSv02 run at any rate in the reference; here the kernels are instantiated for the transform
lengths that 60 ms / 20 ms / 25 ms frames give at 8 ... 48 kHz. Gates: the device's levels equal the CPU oracle's at that rate
bit for bit (the oracle at these rates is pinned against the live binary in tests/test_oracle_rates.py) and, where the real
binary travelled to the box (oracle/_ref), the binary's own levels."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RATES = [8000, 11025, 22050, 32000, 44100, 48000]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture()
def at_rate(oracle):
    def set_rate(fs):
        oracle.set_sample_rate(fs)
    yield set_rate
    oracle.set_sample_rate(16000)


def _utts(fs, seed):
    from opensmile_amd import synth
    lens = [int(2.0 * fs) + 7, int(0.06 * fs) - 1, int(0.06 * fs), int(0.35 * fs), 0, int(1.0 * fs) + 1]
    pcms = [synth.utterance(seed + i, n, fs) if n else np.zeros(0, np.int16) for i, n in enumerate(lens)]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    return pcms, off


@pytest.mark.parametrize("fs", RATES)
def test_f0_chain_levels_at_rate(fs, oracle, at_rate):
    """chain COMPARE_F0: cSpecScale's octave spectrum, cPitchShs's candidates, the 60 ms frame energy and the Viterbi-smoothed
    [F0final, voicing] -- oracle at the same rate, bit for bit."""
    from opensmile_amd import capi
    ctx = capi.Context(0)
    cfg = capi.compare16_f0_config()
    cfg.sample_rate = float(fs)
    plan = capi.Plan(ctx, cfg)
    g = plan.geometry
    assert abs(g.frame_size - 0.06 * fs) <= 0.5 and abs(g.frame_step - 0.01 * fs) <= 0.5       # (22 050 Hz: 1323 and 221 samples)
    pcms, off = _utts(fs, 300 + fs % 13)
    b = capi.Batch(plan, off)
    out, taps = b.f0_run_host_taps(np.concatenate(pcms))
    at_rate(fs)
    oracle.use_reference_fft(False)
    voiced = 0
    for i, p in enumerate(pcms):
        ref, rt = oracle.compare_f0_chain(p, taps=True)
        sl = slice(b.frame_offsets[i], b.frame_offsets[i + 1])
        assert out[sl].shape == ref.shape, (fs, i)
        if not ref.shape[0]:
            continue
        for k in ("hps", "shs", "e60"):
            d = bits(taps[k][sl].reshape(rt[k].shape)) != bits(rt[k])
            assert not d.any(), f"{fs} Hz utt {i} {k}: {d.sum()} of {d.size} words differ, first at {np.argwhere(d)[0]}"
        d = bits(out[sl]) != bits(ref)
        assert not d.any(), f"{fs} Hz utt {i} pitch: {d.sum()} of {d.size} words differ"
        voiced += int((ref[:, 0] > 0).sum())
    assert voiced > 20, "the test signal should have voiced frames"
    b.close()


EG_RATES = [8000, 22050, 32000, 44100, 48000]     # (11 025 Hz: cSpecResample's 20 ms frame gives 221 samples there -- refused)


@pytest.mark.parametrize("fs", EG_RATES)
def test_egemaps_lld_and_functionals_at_rate(fs, oracle, at_rate):
    """The whole eGeMAPSv02 graph (BASELINE config 5) at another rate: 25-column LLD level and 88 functionals equal the oracle's
    at that rate bit for bit (20 ms kernels on FFT 256 / 512 / 1024, F0 group / cHarmonics on FFT 512 ... 4096)."""
    from opensmile_amd import capi, synth
    ctx = capi.Context(0)
    cfg = capi.egemapsv02_config()
    cfg.sample_rate = float(fs)
    plan = capi.Plan(ctx, cfg)
    lens = [int(1.3 * fs) + 5, int(0.06 * fs), int(0.02 * fs) + 1, int(0.5 * fs)]
    pcms = [synth.utterance(500 + i + fs % 11, n, fs) for i, n in enumerate(lens)]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    b = capi.Batch(plan, off)
    lld, func, _ = b.run_host_egemaps(np.concatenate(pcms), taps=True)
    at_rate(fs)
    oracle.use_reference_fft(False)
    fi = 0
    for i, p in enumerate(pcms):
        ref = oracle.egemaps_lld_chain(p)
        got = lld[b.frame_offsets[i]:b.frame_offsets[i + 1]]
        assert got.shape == ref.shape, (fs, i, got.shape, ref.shape)
        d = bits(got) != bits(ref)
        assert not d.any(), f"{fs} Hz utt {i}: {d.sum()} of {d.size} LLD cells differ, columns {sorted(set(np.argwhere(d)[:, 1]))}"
        rf = oracle.egemaps_func(p)
        if rf.shape[0]:
            d = bits(func[i]) != bits(rf[0])
            assert not d.any(), f"{fs} Hz utt {i}: functionals {np.argwhere(d).ravel()[:20]} differ"
            fi += 1
    assert fi >= 2
    b.close()


@pytest.mark.parametrize("fs", RATES)
def test_compare16_lld_and_functionals_at_rate(fs, oracle, at_rate):
    """The whole ComParE_2016 graph (BASELINE config 4) at another rate: the 130-column LLD level equals the oracle's at that rate
    bit for bit (A+B frame kernel on FFT 256 / 512 / 1024, F0 group on FFT 512 ... 4096); the 6373 functionals and the LLD level
    equal the REAL binary's where it travelled to the box."""
    from opensmile_amd import capi, synth
    ctx = capi.Context(0)
    cfg = capi.compare16_config()
    cfg.sample_rate = float(fs)
    plan = capi.Plan(ctx, cfg)
    lens = [int(1.3 * fs) + 5, int(0.06 * fs), int(0.09 * fs) + 1, int(0.5 * fs), int(0.1 * fs)]
    pcms = [synth.utterance(600 + i + fs % 11, n, fs) for i, n in enumerate(lens)]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    b = capi.Batch(plan, off)
    lld, func, _ = b.run_host_with_functionals16(np.concatenate(pcms))
    at_rate(fs)
    oracle.use_reference_fft(False)
    for i, p in enumerate(pcms):
        ref = oracle.compare_lld_chain(p)
        got = lld[b.frame_offsets[i]:b.frame_offsets[i + 1]]
        assert got.shape == ref.shape, (fs, i, got.shape, ref.shape)
        d = bits(got) != bits(ref)
        assert not d.any(), f"{fs} Hz utt {i}: {d.sum()} of {d.size} LLD cells differ, columns {sorted(set(np.argwhere(d)[:, 1]))}"
    if oracle.have_ref():
        for i in (0, 3):
            rf, rl = oracle.run_reference_func("compare16/ComParE_2016.conf", pcms[i], fs=fs)
            got = lld[b.frame_offsets[i]:b.frame_offsets[i + 1]]
            assert got.shape == rl.shape and not (bits(got) != bits(rl)).any(), f"{fs} Hz utt {i}: LLD level differs from the binary's"
            d = bits(func[i]) != bits(rf[0])
            assert not d.any(), f"{fs} Hz utt {i}: {d.sum()} of 6373 functionals differ from the binary's, first {np.argwhere(d).ravel()[:10]}"
    b.close()


@pytest.mark.parametrize("fs", RATES)
def test_is09_lld_at_rate(fs, oracle, at_rate):
    from opensmile_amd import capi, synth
    ctx = capi.Context(0)
    cfg = capi.is09_lld_config()
    cfg.sample_rate = float(fs)
    plan = capi.Plan(ctx, cfg)
    lens = [int(1.1 * fs) + 5, int(0.025 * fs), int(0.3 * fs)]
    pcms = [synth.utterance(700 + i + fs % 11, n, fs) for i, n in enumerate(lens)]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    b = capi.Batch(plan, off)
    out = b.run_host(np.concatenate(pcms))
    at_rate(fs)
    oracle.use_reference_fft(False)
    for i, p in enumerate(pcms):
        ref = oracle.is09_chain(p)
        got = out[b.frame_offsets[i]:b.frame_offsets[i + 1]]
        assert got.shape == ref.shape, (fs, i, got.shape, ref.shape)
        d = bits(got) != bits(ref)
        assert not d.any(), f"{fs} Hz utt {i}: {d.sum()} of {d.size} cells differ, columns {sorted(set(np.argwhere(d)[:, 1]))}"
    b.close()


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "opensmile_amd", "smilextract_hip")
REF = os.path.join(ROOT, "oracle", "_ref", "SMILExtract")
REF_CONF = os.path.join(ROOT, "oracle", "_ref", "config")


@pytest.mark.parametrize("set_name,conf,opts", [
    ("egemapsv02", "egemaps/v02/eGeMAPSv02.conf", ["-lldhtkoutput", "-htkoutput"]),
    ("compare16", "compare16/ComParE_2016.conf", ["-lldhtkoutput", "-htkoutput"]),
    ("is13_compare", "is09-13/IS13_ComParE.conf", ["-lldhtkoutput", "-htkoutput"]),
    ("is09_emotion", "is09-13/IS09_emotion.conf", ["-lldhtkoutput"]),
    ("mfcc12_0_d_a", "mfcc/MFCC12_0_D_A.conf", ["-O"]),
    ("plp_e_d_a_z", "plp/PLP_E_D_A_Z.conf", ["-O"]),
])
def test_smilextract_hip_mixed_rate_list_equals_binary(set_name, conf, opts, tmp_path):
    """One file list holding 8 / 16 / 22.05 / 44.1 / 48 kHz files (one of them 24-bit stereo): one plan per rate, every output
    file equal to the real binary's byte for byte."""
    import subprocess
    import wave
    from opensmile_amd import synth
    from test_gpu_f32_input import write_wav_fmt, _samples
    if not (os.path.exists(REF) and os.path.exists(EXE) and os.path.isdir(REF_CONF)):
        pytest.skip("oracle/_ref/SMILExtract (+ config/) or smilextract_hip not built")
    wavs = []
    for k, fs in enumerate([44100, 8000, 16000, 48000, 22050, 44100]):
        w = str(tmp_path / f"r{k}_{fs}.wav")
        n = int(fs * (1.0 + 0.2 * k)) + k
        if k == 5:
            write_wav_fmt(w, _samples(40 + k, n, 2, 3, 24), fs, 3, 24)
        else:
            with wave.open(w, "wb") as f:
                f.setnchannels(1); f.setsampwidth(2); f.setframerate(fs)
                f.writeframes(synth.utterance(800 + k, n, fs).astype("<i2").tobytes())
        wavs.append(w)
    example = os.path.join(ROOT, "oracle", "_ref", "opensmile.wav")      # the reference's example-audio/opensmile.wav (44.1 kHz)
    if os.path.exists(example):
        wavs.append(example)
    for i, w in enumerate(wavs):
        args = [REF, "-C", os.path.join(REF_CONF, conf), "-I", w, "-l", "0"]
        for o in opts:
            args += [o, str(tmp_path / f"ref{i}{o}.htk")]
        subprocess.run(args, check=True, cwd=str(tmp_path), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lst = tmp_path / "list.txt"
    lst.write_text("".join(w + "\n" for w in wavs))
    outdir = tmp_path / "own"
    outdir.mkdir()
    args = [EXE, "--set", set_name, "-filelist", str(lst), "-outdir", str(outdir)]
    for o in opts:
        args += [o, "on"]
    env = dict(os.environ)
    if not opts[0].startswith("-lld"):
        # cepstral sets: 16-bit mono files at 16 kHz normally take the fast 512-point kernel, whose transform has its own order
        # (within 3e-7 of the reference, tests/test_gpu_mfcc.py); byte identity is the reference-order kernel's property
        env["SMILEHIP_FORCE_GENERIC"] = "1"
    r = subprocess.run(args, capture_output=True, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lld = set_name in ("is09_emotion", "egemapsv02", "compare16", "is13_compare")
    ext = {"-O": ".htk", "-lldhtkoutput": ".lld.htk" if lld else ".htk", "-htkoutput": ".func.htk"}
    for i, w in enumerate(wavs):
        base = os.path.splitext(os.path.basename(w))[0]
        for o in opts:
            ref = open(tmp_path / f"ref{i}{o}.htk", "rb").read()
            own = open(outdir / (base + ext[o]), "rb").read()
            assert len(ref) > 12
            assert own == ref, f"{set_name} {os.path.basename(w)} {o}: files differ ({len(own)} vs {len(ref)} bytes)"
