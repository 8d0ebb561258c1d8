"""SURVEY.md 8(b), "also usable": the reference's in-process C API (progsrc/include/smileapi/SMILEapi.h:73-149, built from the
reference's own SMILEapi.cpp into oracle/_ref/libSMILEapi.so) with the plugin loaded. A host program pushes PCM through
cExternalAudioSource and receives vectors from cExternalSink; the chain between them (MFCC12_0_D_A's components and options,
tests/conf/mfcc_smileapi.conf) is the plugin's overrides when the process starts in a directory with ./plugins. Same vectors
as the same program without the plugin, and the same as the plain binary's file for the utterance."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLUGDIR = os.path.join(ROOT, "opensmile_amd", "plugin")


def _feed(oracle, pcm, cwd, td, tag, env_extra=None):
    lib = os.path.join(oracle.REF_DIR, "libSMILEapi.so")
    plug = os.path.join(PLUGDIR, "plugins", "libsmilehip_plugin.so")
    if not (os.path.exists(lib) and os.path.exists(plug)):
        pytest.skip("oracle/_ref/libSMILEapi.so or the plugin .so not built (needs /root/reference at build time)")
    pin, pout, trace = (os.path.join(td, f"{tag}_{n}") for n in ("pcm.npy", "out.npy", "trace.txt"))
    np.save(pin, pcm)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "opensmile_amd"), oracle.REF_DIR, env.get("LD_LIBRARY_PATH", "")])
    env["SMILEHIP_PLUGIN_TRACE"] = trace
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "smileapi_run.py"), lib,
                        os.path.join(ROOT, "tests", "conf", "mfcc_smileapi.conf"), pin, pout],
                       cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    tr = dict(l.split() for l in open(trace).read().split("\n") if l.strip()) if os.path.exists(trace) else {}
    return np.load(pout), {k: int(v) for k, v in tr.items()}


def test_smileapi_external_source_and_sink_with_plugin(oracle, tmp_path):
    from opensmile_amd import synth
    from tolerance import assert_parity
    td = str(tmp_path)
    pcm = synth.utterance(5, 16000 * 2 + 123)
    plain, tr0 = _feed(oracle, pcm, td, td, "plain")                     # no ./plugins in the working directory
    assert not tr0 and plain.shape == (199, 39), (tr0, plain.shape)
    # the file the plain binary writes for the same samples (the wave source instead of the external one): the same rows
    exe = os.path.join(oracle.REF_DIR, "SMILExtract")
    wav, htk = os.path.join(td, "in.wav"), os.path.join(td, "ref.htk")
    oracle.write_wav(wav, pcm, 16000)
    subprocess.run([exe, "-C", os.path.join(oracle.REF_DIR, "config", "mfcc", "MFCC12_0_D_A.conf"), "-I", wav, "-O", htk, "-l", "0"],
                   cwd=td, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ref = oracle.read_htk(htk)[0]
    assert np.array_equal(plain.view(np.uint32), ref.view(np.uint32))
    own, tr = _feed(oracle, pcm, PLUGDIR, td, "plug")                    # ./plugins/libsmilehip_plugin.so takes part
    T = 199
    assert tr.get("cMfcc", 0) == T and tr.get("cTransformFFT", 0) == T and tr.get("cDeltaRegression", 0) >= 2 * T, tr
    assert not [k for k, v in tr.items() if k.endswith(".cpu") and v], tr
    assert own.shape == plain.shape
    assert_parity(own, plain.astype(np.float64), block=13, what="SMILEapi + plugin vs SMILEapi alone")
