"""The F0 chunk pipeline (lld_launch.hpp: F0Pipe -- chunk i's candidates beside chunk i + 1's sweep and chunk i + 2's spectra, two sets
of scratch rows): the whole ComParE_2016 and eGeMAPSv02 LLD levels of a batch cut into MANY chunks (SMILEHIP_F0_CHUNK_TILES=64) equal,
bit for bit, the same batch run chunk after chunk on one stream (the default; the pipeline is SMILEHIP_F0_PIPE=1 -- measured no
faster) and as a single chunk. The switches are read once
per process, so every variant is a process of its own."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from opensmile_amd import capi, synth
cfg = getattr(capi, sys.argv[1])()
ctx = capi.Context(0)
plan = capi.Plan(ctx, cfg)
lens = [16000 * 3 + 137 * (i %% 7) for i in range(96)]
pcms = [synth.utterance(2 + (i %% 9), n) for i, n in enumerate(lens)]
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
b = capi.Batch(plan, off)
out = b.run_host(np.concatenate(pcms))
out2 = b.run_host(np.concatenate(pcms))          # a second run of the same batch: the events and the scratch sets are taken again
assert np.array_equal(out.view(np.uint32), out2.view(np.uint32))
np.save(sys.argv[2], out)
""" % ROOT


def run(cfg, path, **env):
    e = dict(os.environ, **env)
    p = subprocess.run([sys.executable, "-c", SCRIPT, cfg, path], env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return np.load(path)


@pytest.mark.parametrize("cfg", ["compare16_config", "egemapsv02_config"])
def test_chunk_pipeline_bit_identical(cfg, tmp_path):
    one = run(cfg, str(tmp_path / "one.npy"))                                                     # a single chunk (no pipeline)
    piped = run(cfg, str(tmp_path / "piped.npy"), SMILEHIP_F0_CHUNK_TILES="64", SMILEHIP_F0_PIPE="1")   # ~60 chunks, pipelined
    serial = run(cfg, str(tmp_path / "serial.npy"), SMILEHIP_F0_CHUNK_TILES="64", SMILEHIP_F0_PIPE="0")
    assert one.shape == piped.shape == serial.shape and one.shape[0] > 20000
    assert np.array_equal(piped.view(np.uint32), serial.view(np.uint32))
    assert np.array_equal(piped.view(np.uint32), one.view(np.uint32))
