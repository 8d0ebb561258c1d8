"""Float-sample input of the fused chains (smilehip_lld_run_f32), every integer sample format and IEEE-float files through smilextract_hip.

cWaveSource hands the graph floats whatever the file held (smilePcm_convertSamples, smileUtil.c:2500-2627, monoMixdown = 1 in
config/shared/standard_wave_input.conf.inc); the 16-bit mono run converts at the kernels' loads. The two must agree bit for bit:
(1) run_device_f32 on smilehip_pcm16_to_float's output == run_device on the same int16 batch, every chain; (2) 8 / 24 / 32-bit
and multi-channel WAV files through smilextract_hip == the REAL binary's files, byte for byte (oracle/_ref/SMILExtract, which
travels to the GPU box with its config/ directory)."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "opensmile_amd", "smilextract_hip")
REF = os.path.join(ROOT, "oracle", "_ref", "SMILExtract")
REF_CONF = os.path.join(ROOT, "oracle", "_ref", "config")


def _batch(seed, lens):
    from opensmile_amd import synth
    pcms = [synth.utterance(seed + i, n) if n else np.zeros(0, np.int16) for i, n in enumerate(lens)]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    return np.concatenate(pcms), off


def _configs():
    from opensmile_amd import capi
    return {"mfcc12_0_d_a": capi.mfcc12_0_d_a_config, "plp_0_d_a": capi.plp_0_d_a_config, "is09": capi.is09_lld_config,
            "compare16": capi.compare16_config, "is13": capi.is13_compare_config, "egemapsv02": capi.egemapsv02_config,
            "mfcc12_e_d_a_z": lambda: capi.htk_variant_config("MFCC12_E_D_A_Z")}


@pytest.mark.parametrize("name", ["mfcc12_0_d_a", "plp_0_d_a", "mfcc12_e_d_a_z", "is09", "compare16", "is13", "egemapsv02"])
@pytest.mark.parametrize("odd", [0, 1])
def test_float_input_equals_int16_input_bit_for_bit(name, odd, monkeypatch):
    import torch
    from opensmile_amd import capi
    ctx = capi.Context(0)
    plan = capi.Plan(ctx, _configs()[name]())                  # (cepstral sets: the plan with the fast 512-point kernel)
    plan16 = plan
    if name in ("mfcc12_0_d_a", "plp_0_d_a", "mfcc12_e_d_a_z"):
        # float input runs the reference-order kernel; the int16 run to hold it against is that kernel too (the fast kernel
        # has its own transform order: within 3e-7 of it, tests/test_gpu_mfcc.py, not bit for bit)
        monkeypatch.setenv("SMILEHIP_FORCE_GENERIC", "1")
        plan16 = capi.Plan(ctx, _configs()[name]())
    lens = [16000 * 3 + odd, 1234 + odd, 16000, 399, 0, 52001] if name != "egemapsv02" else [16000 * 3 + odd, 9601, 16000 + odd, 52001]
    pcm, off = _batch(700, lens)
    b = capi.Batch(plan, off)
    n_out = plan.geometry.n_out
    d_pcm = torch.from_numpy(np.concatenate([pcm, np.zeros(2, np.int16)])).cuda()
    d_f32 = torch.empty(len(pcm) + 2, dtype=torch.float32, device="cuda")
    capi._check(capi.load().smilehip_pcm16_to_float(ctx._h, d_pcm.data_ptr(), len(pcm), d_f32.data_ptr(), None))
    d_a = torch.full((max(b.total_rows, 1), n_out), float("nan"), dtype=torch.float32, device="cuda")
    d_b = torch.full((max(b.total_rows, 1), n_out), float("nan"), dtype=torch.float32, device="cuda")
    (b if plan16 is plan else capi.Batch(plan16, off)).run_device(d_pcm.data_ptr(), d_a.data_ptr(), n_out)
    torch.cuda.synchronize()
    b.run_device_f32(d_f32.data_ptr(), d_b.data_ptr(), n_out)
    torch.cuda.synchronize()
    a, f = d_a.cpu().numpy().view(np.uint32), d_b.cpu().numpy().view(np.uint32)
    assert b.total_rows > 0
    diff = a != f
    assert not diff.any(), f"{name}: {diff.sum()} of {diff.size} words differ, columns {sorted(set(np.argwhere(diff)[:, 1]))[:20]}"


def write_wav_fmt(path, x, fs, n_bps, n_bits=None):
    """x: int array (n, n_chan) of sample values already in the target range."""
    n_bits = n_bits or 8 * n_bps
    x = np.asarray(x)
    n, ch = x.shape
    if n_bps == 1:
        data = x.astype(np.int8).tobytes()                      # the reference reads 8-bit samples as signed chars
    elif n_bps == 2:
        data = x.astype("<i2").tobytes()
    elif n_bps == 3:
        v = x.astype("<i4").reshape(-1)
        data = np.stack([(v & 0xff), (v >> 8) & 0xff, (v >> 16) & 0xff], axis=1).astype(np.uint8).tobytes()
    else:
        data = x.astype("<i4").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, ch, fs, fs * ch * n_bps, ch * n_bps, n_bits))
        f.write(b"data" + struct.pack("<I", len(data)) + data)


def _samples(seed, n, ch, n_bps, n_bits):
    from opensmile_amd import synth
    full = {1: 127, 2: 32767, 3: 32767 * 256, 4: (32767 * 256 if n_bits == 24 else 2147483647)}[n_bps]
    cols = []
    for c in range(ch):
        s = synth.utterance(seed + 17 * c, n).astype(np.float64) / 32768.0
        cols.append(np.clip(np.round(s * full * (0.9 - 0.2 * c)), -full, full).astype(np.int64))
    return np.stack(cols, axis=1)


FORMATS = [(2, 16, 2), (3, 24, 1), (1, 8, 1), (4, 32, 2), (4, 24, 1), (2, 16, 3), (3, 24, 2)]


@pytest.mark.parametrize("set_name,conf,opts", [
    ("mfcc12_0_d_a", "mfcc/MFCC12_0_D_A.conf", ["-O"]),
    ("is09_emotion", "is09-13/IS09_emotion.conf", ["-lldhtkoutput"]),
    ("egemapsv02", "egemaps/v02/eGeMAPSv02.conf", ["-lldhtkoutput", "-htkoutput"]),
    ("compare16", "compare16/ComParE_2016.conf", ["-lldhtkoutput", "-htkoutput"]),
])
def test_smilextract_hip_every_integer_format_equals_binary(set_name, conf, opts, tmp_path):
    if not (os.path.exists(REF) and os.path.exists(EXE) and os.path.isdir(REF_CONF)):
        pytest.skip("oracle/_ref/SMILExtract (+ config/) or smilextract_hip not built")
    wavs = []
    for k, (n_bps, n_bits, ch) in enumerate(FORMATS):
        w = str(tmp_path / f"f{k}.wav")
        write_wav_fmt(w, _samples(900 + k, 16000 * 2 + 37 * k, ch, n_bps, n_bits), 16000, n_bps, n_bits)
        wavs.append(w)
    w = str(tmp_path / "plain.wav")                                 # a 16-bit mono file in the same batch (mixed batch -> float path)
    write_wav_fmt(w, _samples(950, 16000 + 5, 1, 2, 16), 16000, 2, 16)
    wavs.append(w)
    # the reference, one process per file
    for i, w in enumerate(wavs):
        args = [REF, "-C", os.path.join(REF_CONF, conf), "-I", w, "-l", "0"]
        for o in opts:
            args += [o, str(tmp_path / f"ref{i}{o}.htk")]
        subprocess.run(args, check=True, cwd=str(tmp_path), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # ours, one batch over the list (per-file outputs of a list: -outdir/<basename><ext>)
    lst = tmp_path / "list.txt"
    lst.write_text("".join(w + "\n" for w in wavs))
    outdir = tmp_path / "own"
    outdir.mkdir()
    args = [EXE, "--set", set_name, "-filelist", str(lst), "-outdir", str(outdir)]
    for o in opts:
        args += [o, "on"]
    r = subprocess.run(args, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lld = set_name in ("is09_emotion", "egemapsv02", "compare16")
    ext = {"-O": ".htk", "-lldhtkoutput": ".lld.htk" if lld else ".htk", "-htkoutput": ".func.htk"}
    for i, w in enumerate(wavs):
        base = os.path.splitext(os.path.basename(w))[0]
        for o in opts:
            ref = open(tmp_path / f"ref{i}{o}.htk", "rb").read()
            own = open(outdir / (base + ext[o]), "rb").read()
            assert len(ref) > 12
            assert own == ref, f"{set_name} file {i} ({FORMATS[i] if i < len(FORMATS) else 'plain'}) {o}: files differ"


def write_wav_float(path, x, fs):
    """x: float32 array (n, n_chan): WAVE_FORMAT_IEEE_FLOAT (3), 32 bit"""
    x = np.asarray(x, "<f4")
    n, ch = x.shape
    data = x.tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 3, ch, fs, fs * ch * 4, ch * 4, 32))
        f.write(b"data" + struct.pack("<I", len(data)) + data)


@pytest.mark.parametrize("set_name,conf,opts", [
    ("mfcc12_0_d_a", "mfcc/MFCC12_0_D_A.conf", ["-O"]),
    ("egemapsv02", "egemaps/v02/eGeMAPSv02.conf", ["-lldhtkoutput", "-htkoutput"]),
])
def test_smilextract_hip_ieee_float_wav_equals_binary(set_name, conf, opts, tmp_path):
    """32-bit IEEE float files (smilePcm_convertFloatSamples, smileUtil.c:2629-2690): mono, stereo and three channels (the
    mix-down's float additions in channel order), values beyond +-1 and a negative zero included; in one batch with an
    integer file."""
    if not (os.path.exists(REF) and os.path.exists(EXE) and os.path.isdir(REF_CONF)):
        pytest.skip("oracle/_ref/SMILExtract (+ config/) or smilextract_hip not built")
    from opensmile_amd import synth
    wavs = []
    for k, ch in enumerate([1, 2, 3]):
        n = 16000 * 2 + 41 * k
        x = np.stack([synth.utterance(960 + k + 13 * c, n).astype(np.float32) / np.float32(32768.0) * np.float32(1.3 - 0.4 * c)
                      for c in range(ch)], axis=1)
        x[5, :] = -0.0
        w = str(tmp_path / f"fl{k}.wav")
        write_wav_float(w, x, 16000)
        wavs.append(w)
    w = str(tmp_path / "plain.wav")
    write_wav_fmt(w, _samples(970, 16000 + 5, 1, 2, 16), 16000, 2, 16)
    wavs.append(w)
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
    r = subprocess.run(args, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lld = set_name != "mfcc12_0_d_a"
    ext = {"-O": ".htk", "-lldhtkoutput": ".lld.htk" if lld else ".htk", "-htkoutput": ".func.htk"}
    for i, w in enumerate(wavs):
        base = os.path.splitext(os.path.basename(w))[0]
        for o in opts:
            ref = open(tmp_path / f"ref{i}{o}.htk", "rb").read()
            own = open(outdir / (base + ext[o]), "rb").read()
            assert len(ref) > 12
            assert own == ref, f"{set_name} file {i} {o}: files differ"


def test_smilextract_hip_refuses_other_float_widths(tmp_path):
    """64-bit float samples: the reference prints 'cannot convert unknown sample format' and delivers nothing; refused by name here"""
    w = str(tmp_path / "f.wav")
    data = np.zeros(1600, "<f8").tobytes()
    with open(w, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 3, 1, 16000, 128000, 8, 64))
        f.write(b"data" + struct.pack("<I", len(data)) + data)
    r = subprocess.run([EXE, "--set", "mfcc12_0_d_a", "-I", w, "-O", str(tmp_path / "o.htk")], capture_output=True)
    assert r.returncode != 0 and b"IEEE float" in r.stderr
