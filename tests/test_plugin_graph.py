"""The plugin recognises the graph from the cConfigManager the component loader hands to registerPluginComponent
(src/include/core/componentManager.hpp:23, src/include/core/configManager.hpp:567-650) -- not from the process's command line, not
by reading the file a second time. Runs WITHOUT a GPU: recognition happens before the first device call, which then fails loudly
(the library has no CPU path), so the log of the unmodified reference binary shows both facts."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLUGDIR = os.path.join(ROOT, "opensmile_amd", "plugin")
REF = os.path.join(ROOT, "oracle", "_ref")
EXE = os.path.join(REF, "SMILExtract")
PLUG = os.path.join(PLUGDIR, "plugins", "libsmilehip_plugin.so")
WAV = os.path.join(ROOT, "tests", "golden", "files", "u3_4000.wav")

needs_ref = pytest.mark.skipif(not (os.path.exists(EXE) and os.path.exists(PLUG)), reason="oracle/_ref/SMILExtract or the plugin .so not built")


def run(conf, *extra, env_extra=None):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "opensmile_amd"), REF, env.get("LD_LIBRARY_PATH", "")])
    env.pop("SMILEHIP_PLUGIN_FUSE", None)
    env.update(env_extra or {})
    r = subprocess.run([EXE, "-C", os.path.join(REF, "config", conf), "-I", WAV, "-l", "3", *extra], cwd=PLUGDIR, env=env, capture_output=True,
                       text=True, errors="replace", timeout=120)
    return r.returncode, r.stdout + r.stderr


def gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@needs_ref
@pytest.mark.parametrize("conf,what", [
    ("mfcc/MFCC12_0_D_A.conf", "MFCC chain: 25 ms / 10 ms frames, 26 bands 0-8000 Hz, 13 cepstra, 2 delta stage(s)"),
    ("plp/PLP_E_D_A_Z.conf", "PLP chain"),
    ("is09-13/IS09_emotion.conf", "the graph of IS09_emotion.conf"),
    ("compare16/ComParE_2016.conf", "the graph of ComParE_2016.conf"),
    ("is09-13/IS13_ComParE.conf", "the graph of IS13_ComParE.conf"),
    ("egemaps/v02/eGeMAPSv02.conf", "the graph of eGeMAPSv02.conf"),
])
def test_graph_recognised_from_the_config_manager(conf, what, tmp_path):
    if gpu_present():
        pytest.skip("this test reads the log of a run that stops at the first device call")
    rc, log = run(conf, "-O", str(tmp_path / "o.out"))
    assert "libsmilehip plugin: fused mode -- " + what in log, log[-1500:]
    # ... and then nothing is computed anywhere else: the first device call is an error of the process
    assert "no HIP device visible (libsmilehip has no CPU fallback)" in log, log[-1500:]


@needs_ref
def test_command_line_options_of_the_file_reach_the_plan(tmp_path):
    """The \\cm[...] options a file defines are answered by the loader's own command-line parser: IS09's F0 parameters are such options
    in none of the shipped files, but the output options are -- a run with output options set differently is still the same graph."""
    if gpu_present():
        pytest.skip("this test reads the log of a run that stops at the first device call")
    rc, log = run("is09-13/IS09_emotion.conf", "-lldhtkoutput", str(tmp_path / "l.htk"), "-instname", "x", "-O", str(tmp_path / "o.arff"))
    assert "libsmilehip plugin: fused mode -- the graph of IS09_emotion.conf" in log, log[-1500:]


@needs_ref
def test_unrecognised_graph_and_partial_override_lists_take_the_block_path(tmp_path):
    if gpu_present():
        pytest.skip("this test reads the log of a run that stops at the first device call")
    rc, log = run("prosody/prosodyShs.conf", "-O", str(tmp_path / "o.csv"), env_extra={"SMILEHIP_PLUGIN_FUSE": "1"})
    assert "fused mode -- " not in log.replace("fused mode: ", "") and "block-per-tick path" in log, log[-1500:]
    rc, log = run("mfcc/MFCC12_0_D_A.conf", "-O", str(tmp_path / "o.htk"), env_extra={"SMILEHIP_PLUGIN_FUSE": "1", "SMILEHIP_PLUGIN_COMPONENTS": "cMfcc,cMelspec"})
    assert "fused mode needs every override registered" in log, log[-1500:]


@needs_ref
def test_no_side_channel_in_the_plugin_sources():
    for fn in os.listdir(PLUGDIR):
        if fn.endswith((".hpp", ".cpp")):
            txt = open(os.path.join(PLUGDIR, fn)).read()
            assert "cmdline" not in txt and "/proc/" not in txt, fn
    assert b"/proc/self" not in open(PLUG, "rb").read()
    assert b"SMILEHIP_PLUGIN_ALLOW_CPU" not in open(PLUG, "rb").read()
