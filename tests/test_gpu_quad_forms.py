"""The sixteen-lanes-per-frame kernels of round 5 (lld_compare_quad.hpp, lld_gemaps_quad.hpp) against the wave-per-frame forms they
replace, through the C ABI, bit for bit: ragged batches with the edge cases of the run tables -- utterances without a frame, with
one frame, shorter than the 60 ms window, frame counts around the run length, a run count that fills neither a wave's four rows nor
a workgroup's sixteen --, every run length the batch may pick (1 frame per run: every frame is its own warm-up; 3; 8; 64), and the
all-zero and clipping utterances of the corpus contract."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LENS = [160000, 0, 100, 319, 320, 479, 480, 959, 960, 961, 1119, 1120, 1439, 1440, 1600, 1760, 2720, 9000, 8720, 21280, 48000, 11000,
        320 + 160 * 7, 320 + 160 * 8, 320 + 160 * 9, 320 + 160 * 63, 320 + 160 * 64, 320 + 160 * 65, 33333]


def _corpus():
    from opensmile_amd import synth
    pcm = np.concatenate([synth.utterance(i % 13, n) if n else np.zeros(0, np.int16) for i, n in enumerate(LENS)])   # (0: zeros, 1: clipping square, 10: noise)
    off = np.concatenate([[0], np.cumsum(LENS)]).astype(np.int64)
    return pcm, off


@pytest.mark.parametrize("run_frames", [None, "1", "3", "8", "64"])
def test_compare_quad_form_equals_wave_form_bit_for_bit(monkeypatch, run_frames):
    from opensmile_amd import capi
    pcm, off = _corpus()
    ctx = capi.Context(0)
    plan = capi.Plan(ctx, capi.compare16_config())
    outs = {}
    for form in ("quad", "wave"):
        monkeypatch.delenv("SMILEHIP_COMPARE_WAVE", raising=False)
        monkeypatch.delenv("SMILEHIP_RUN_FRAMES", raising=False)
        if form == "wave":
            monkeypatch.setenv("SMILEHIP_COMPARE_WAVE", "1")
        if run_frames:
            monkeypatch.setenv("SMILEHIP_RUN_FRAMES", run_frames)
        b = capi.Batch(plan, off)
        outs[form] = b.run_host(pcm).copy()
        b.close()
    assert outs["quad"].shape == outs["wave"].shape and outs["quad"].shape[1] == 130 and outs["quad"].shape[0] > 2000
    d = outs["quad"].view(np.uint32) != outs["wave"].view(np.uint32)
    assert not d.any(), (int(d.sum()), sorted(set(np.argwhere(d)[:, 1]))[:20], np.argwhere(d)[:5])
    plan.close()
    ctx.close()


@pytest.mark.parametrize("run_frames", [None, "1", "5", "64"])
def test_gemaps_quad_form_equals_wave_form_bit_for_bit(monkeypatch, run_frames):
    from opensmile_amd import capi
    pcm, off = _corpus()
    ctx = capi.Context(0)
    for cfg_fn in (capi.egemapsv02_config, capi.egemapsv01a_config):      # (96 zeros in front of the frame / none: the two instances)
        plan = capi.Plan(ctx, cfg_fn())
        outs = {}
        for form in ("quad", "wave"):
            monkeypatch.delenv("SMILEHIP_GEMAPS_WAVE", raising=False)
            monkeypatch.delenv("SMILEHIP_RUN_FRAMES", raising=False)
            if form == "wave":
                monkeypatch.setenv("SMILEHIP_GEMAPS_WAVE", "1")
            if run_frames:
                monkeypatch.setenv("SMILEHIP_RUN_FRAMES", run_frames)
            b = capi.Batch(plan, off)
            lld, func, taps = b.run_host_egemaps(pcm, taps=True)
            outs[form] = (lld.copy(), func.copy(), taps["raw20"].copy(), taps["lpc"].copy(), taps["formants"].copy())
            b.close()
        for i, what in enumerate(("LLD level", "functionals", "20 ms raw descriptors", "LP coefficients", "formants")):
            a, w = outs["quad"][i], outs["wave"][i]
            assert a.shape == w.shape and a.size > 0, what
            d = a.view(np.uint32) != w.view(np.uint32)
            assert not d.any(), (what, int(d.sum()), np.argwhere(d)[:5])
        plan.close()
    ctx.close()


def test_harmonics_from_kept_magnitudes_equal_recomputed_ones_bit_for_bit(monkeypatch):
    """cHarmonics reads the level cSpecScale reads: lld_gemaps_harm on the spectra lld_f0_spec kept (the default when they fit) against
    lld_gemaps_harm transforming the frames itself (SMILEHIP_HARM_KEEP_MAG=0)."""
    from opensmile_amd import capi
    pcm, off = _corpus()
    ctx = capi.Context(0)
    for cfg_fn in (capi.egemapsv02_config, capi.egemapsv01a_config):
        plan = capi.Plan(ctx, cfg_fn())
        outs = {}
        for keep in ("1", "0"):
            monkeypatch.setenv("SMILEHIP_HARM_KEEP_MAG", keep)
            b = capi.Batch(plan, off)
            lld, func, taps = b.run_host_egemaps(pcm, taps=True)
            lld2, func2, taps2 = b.run_host_egemaps(pcm, taps=True)       # (a second run of the same batch: counters and stores reset)
            assert np.array_equal(lld.view(np.uint32), lld2.view(np.uint32)) and np.array_equal(func.view(np.uint32), func2.view(np.uint32))
            outs[keep] = (lld.copy(), func.copy(), taps["harm6"].copy())
            b.close()
        assert outs["1"][2].shape[1] == 6 and (outs["1"][2][:, 0] != 0).any()            # (voiced frames: an HNR)
        for i, what in enumerate(("LLD level", "functionals", "cHarmonics' six outputs")):
            a, w = outs["1"][i], outs["0"][i]
            assert a.shape == w.shape and a.size > 0, what
            d = a.view(np.uint32) != w.view(np.uint32)
            assert not d.any(), (what, int(d.sum()), np.argwhere(d)[:5])
        plan.close()
    ctx.close()
