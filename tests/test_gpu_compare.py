"""Config-3 shape, second set: ComParE_2016's LLD groups A and B (59 LLD + 59 delta,
T60+1 rows) through the C ABI's smilehip_lld_run, against golden outputs of the real
reference binary and against the CPU oracle. Rows R8 (cPlp auditory spectrum, with and
without RASTA), R11 (cSpectral), R12 on the 20 ms / 60 ms frames, SMA+delta over levels
of different lengths (R13)."""
import numpy as np
import pytest

from test_oracle_pin_compare import KEYS, compare_tolerances

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from opensmile_amd import capi
    ctx = capi.Context(0)
    plan = capi.Plan(ctx, capi.compare16_ab_config())
    assert (plan.geometry.n_static, plan.geometry.n_out) == (59, 118)
    return capi, ctx, plan


def test_compare_ab_golden_batch_ragged(hip, golden_compare):
    capi, ctx, plan = hip
    pcms = [golden_compare["pcm_" + k] for k in KEYS]
    refs = [golden_compare["out_" + k] for k in KEYS]
    off = np.concatenate([[0], np.cumsum([len(p) for p in pcms])]).astype(np.int64)
    b = capi.Batch(plan, off)
    np.testing.assert_array_equal(np.diff(b.frame_offsets), [r.shape[0] for r in refs])   # T60+1 rows
    out = b.run_host(np.concatenate(pcms))
    for i, k in enumerate(KEYS):
        o = out[b.frame_offsets[i]:b.frame_offsets[i + 1]]
        compare_tolerances(o, refs[i], k)
        # zcr (col 3) is integer work: exact
        assert np.array_equal(o[:, 3], refs[i][:, 3]), k
    b.close()


def test_compare_ab_vs_oracle_ragged_lengths(hip, oracle):
    """10 s utterances and every short length class (no rows below four 60 ms frames,
    the tick-accurate short chain up to 16 frames, tile boundaries of the window chain)."""
    capi, ctx, plan = hip
    from opensmile_amd import synth
    lens = [160000, 100, 959, 960, 1439, 1440, 1600, 1760, 2720, 2880, 3040, 21120, 21280, 21440, 160000, 48000, 0]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pcm = np.concatenate([synth.utterance(50 + i, n) if n else np.zeros(0, np.int16) for i, n in enumerate(lens)])
    b = capi.Batch(plan, off)
    out = b.run_host(pcm)
    assert out.shape[1] == 118
    for i, n in enumerate(lens):
        ref = oracle.compare_ab_chain(pcm[off[i]:off[i + 1]])
        o = out[b.frame_offsets[i]:b.frame_offsets[i + 1]]
        assert o.shape == ref.shape, (n, o.shape, ref.shape)
        if ref.shape[0]:
            compare_tolerances(o, ref, f"len{n}")
    b.close()


def test_compare_ab_rerun_is_deterministic(hip):
    capi, ctx, plan = hip
    from opensmile_amd import synth
    pcm = np.concatenate([synth.utterance(3, 32000), synth.utterance(4, 16000)])
    b = capi.Batch(plan, np.array([0, 32000, 48000], dtype=np.int64))
    a1 = b.run_host(pcm)
    a2 = b.run_host(pcm)
    assert np.array_equal(a1.view(np.uint32), a2.view(np.uint32))
    b.close()


def test_compare_wave_kernel_equals_block_kernel(tmp_path):
    """The wave-per-frame frame kernel keeps the summation order of the workgroup-per-run kernel (SMILEHIP_COMPARE_BLOCK=1
    selects the latter at its first launch, hence two processes): bit-identical rows."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import torch\n"
        "from opensmile_amd import capi, synth\n"
        "ctx = capi.Context(0); plan = capi.Plan(ctx, capi.compare16_ab_config())\n"
        "lens = [48000, 1600, 9000, 160000, 2720]\n"
        "off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)\n"
        "pcm = np.concatenate([synth.utterance(40 + i, n) for i, n in enumerate(lens)])\n"
        "b = capi.Batch(plan, off); np.save(sys.argv[1], b.run_host(pcm))\n" % root)
    outs = []
    for tag, env_extra in (("wave", {}), ("block", {"SMILEHIP_COMPARE_BLOCK": "1"})):
        path = str(tmp_path / (tag + ".npy"))
        env = dict(os.environ)
        env.update(env_extra)
        subprocess.run([sys.executable, "-c", code, path], check=True, env=env)
        outs.append(np.load(path))
    assert outs[0].shape == outs[1].shape and outs[0].shape[1] == 118
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
