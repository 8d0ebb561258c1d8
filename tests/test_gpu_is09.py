"""Config-3 shape: IS09_emotion's LLD level (16 LLD + 16 delta, T+1 rows) through the C
ABI's smilehip_lld_run, against golden outputs of the real reference binary and against
the CPU oracle. Rows R9 (cAcf), R10 (cPitchACF), R12 (cEnergy, cMZcr), SMA+delta of R13."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ["u2_16000", "u3_16000", "u10_16000", "u1_16000", "u0_16000", "u7_399", "u7_400", "u7_560",
        "u7_720", "u7_880", "u7_1040", "u4_48000"]


@pytest.fixture(scope="module")
def hip():
    from opensmile_amd import capi
    ctx = capi.Context(0)
    plan = capi.Plan(ctx, capi.is09_lld_config())
    assert (plan.geometry.n_static, plan.geometry.n_out) == (16, 32)
    return capi, ctx, plan


def check_lld(out, ref, what):
    """IS09_emotion's 16 LLD + 16 deltas against the real binary / the oracle: identical bits -- MFCC 1-12 (rdft network +
    glibc logf), RMS energy, ZCR, cAcf's ACF and cepstrum (inverse rdft network), cPitchACF's voicing probability and F0 with its
    smoother. Round 2 gated 1e-5 per column and one F0 flip event per input."""
    from tolerance import assert_bits_equal
    assert np.isfinite(out).all()
    assert_bits_equal(out, ref, what)
    return 0.0


def test_is09_golden_batch_ragged(hip, golden_is09):
    capi, ctx, plan = hip
    pcms = [golden_is09["pcm_" + k] for k in KEYS]
    refs = [golden_is09["out_" + k] for k in KEYS]
    off = np.concatenate([[0], np.cumsum([len(p) for p in pcms])]).astype(np.int64)
    b = capi.Batch(plan, off)
    # integer contract: T+1 rows per non-empty utterance
    np.testing.assert_array_equal(np.diff(b.frame_offsets), [r.shape[0] for r in refs])
    out = b.run_host(np.concatenate(pcms))
    worst = 0.0
    for i, k in enumerate(KEYS):
        o = out[b.frame_offsets[i]:b.frame_offsets[i + 1]]
        if refs[i].shape[0] == 0:
            assert o.shape[0] == 0
            continue
        worst = max(worst, check_lld(o, refs[i], k))
    print(f"worst F0 flip rate {worst * 100:.2f}%")
    b.close()


def test_is09_vs_oracle_10s_and_smoothing_chain_exact(hip, oracle):
    """10 s utterances vs the oracle; and, given the GPU's own pre-SMA columns, the
    SMA+delta chain must equal the oracle's tick-accurate chain bit for bit."""
    capi, ctx, plan = hip
    import ctypes as C
    from opensmile_amd import synth
    us = [5, 20, 31]
    pcms = [synth.utterance(u, 160000) for u in us]
    off = np.arange(len(us) + 1, dtype=np.int64) * 160000
    b = capi.Batch(plan, off)
    out = b.run_host(np.concatenate(pcms))
    for i, u in enumerate(us):
        ref = oracle.is09_chain(pcms[i])
        assert ref.shape == (999, 32)
        check_lld(out[b.frame_offsets[i]:b.frame_offsets[i + 1]], ref, f"u{u}")
    b.close()


def test_window_chain_sma_delta_bit_exact(hip, oracle):
    """The generic window chain on arbitrary data: SMA(3) -> delta(2), every length
    1..20 and tile boundaries, bit-exact vs the oracle's tick-accurate simulation.
    Driven through the IS09 path with a hand-made scratch matrix is not possible via
    the public ABI, so this uses the delta entry point for the delta->delta chain and
    relies on test_is09_* for SMA->delta: here only the row bookkeeping is checked."""
    capi, ctx, plan = hip
    lens = [400 + 160 * (T - 1) for T in (1, 2, 3, 4, 5, 16, 17, 127, 128, 129, 300)]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    b = capi.Batch(plan, off)
    assert list(np.diff(b.frame_offsets)) == [T + 1 for T in (1, 2, 3, 4, 5, 16, 17, 127, 128, 129, 300)]
    from opensmile_amd import synth
    pcm = np.concatenate([synth.utterance(40 + i, n) for i, n in enumerate(lens)])
    out = b.run_host(pcm)
    for i, n in enumerate(lens):
        ref = oracle.is09_chain(pcm[off[i]:off[i + 1]])
        check_lld(out[b.frame_offsets[i]:b.frame_offsets[i + 1]], ref, f"len{n}")
    b.close()


def test_is09_other_sample_rates_wave_equals_workgroup_kernel():
    """Geometries other than 16 kHz / 25 ms: 8 kHz (N = 200, M = 128: the wave kernel's run-time-M path with the generic pair
    transform), 32 kHz (N = 800, M = 512: the fused transform) -- the wave-per-frame kernel against the one-workgroup-per-frame
    kernel (SMILEHIP_IS09=block), which runs the in-place radix-2 transform and the workgroup reductions: same values within
    the chain's tolerance (the energy sums use different trees)."""
    import os
    from opensmile_amd import capi, synth
    ctx = capi.Context(0)
    for fs in (8000, 32000):
        cfg = capi.is09_lld_config()
        cfg.sample_rate = float(fs)
        plan = capi.Plan(ctx, cfg)
        lens = [3 * fs, fs // 2, fs]
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        pcm = np.concatenate([synth.utterance(40 + i, n, fs=fs) for i, n in enumerate(lens)])
        b = capi.Batch(plan, off)
        a = b.run_host(pcm)
        os.environ["SMILEHIP_IS09"] = "block"
        try:
            r = b.run_host(pcm)
        finally:
            del os.environ["SMILEHIP_IS09"]
        b.close()
        assert a.shape == r.shape and a.shape[0] > 0 and np.isfinite(a).all()
        check_lld(a, r.astype(np.float64), f"fs{fs}")


def test_is09_quad_form_equals_wave_form_bit_for_bit(monkeypatch):
    """16 kHz / 25 ms: the sixteen-lanes-per-frame kernel (four frames per wave, lld_ooura_quad.hpp) against the wave-per-frame
    kernel (SMILEHIP_IS09=wave) -- the same butterflies on the same operands, the same sequential band / cepstrum sums: every
    cell but the RMS energy (a double sum whose association differs; rounded to float afterwards) must carry the same bits, and
    the energy column too on these inputs. Ragged lengths: a batch whose last pass is not full, the all-zero and the clipping
    utterance."""
    from opensmile_amd import capi, synth
    ctx = capi.Context(0)
    plan = capi.Plan(ctx, capi.is09_lld_config())
    _quad_vs_wave(capi, synth, plan, 400, monkeypatch)
    cfg = capi.is09_lld_config()                           # 32 ms frames: 512 samples, still FFT 512 (32 samples per lane)
    cfg.frame_size_sec = 0.032
    _quad_vs_wave(capi, synth, capi.Plan(ctx, cfg), 512, monkeypatch)


def _quad_vs_wave(capi, synth, plan, N, monkeypatch):
    lens = [48000, N, N + 161, 16000, 160000, 24000, N - 1, N + 319]
    seeds = [0, 1, 2, 3, 4, 10, 5, 6]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pcm = np.concatenate([synth.utterance(s, n) for s, n in zip(seeds, lens)])
    b = capi.Batch(plan, off)
    monkeypatch.delenv("SMILEHIP_IS09", raising=False)
    q = b.run_host(pcm)
    monkeypatch.setenv("SMILEHIP_IS09", "wave")
    w = b.run_host(pcm)
    monkeypatch.delenv("SMILEHIP_IS09", raising=False)
    b.close()
    assert q.shape == w.shape and q.shape[0] == sum(max(0, (n - N) // 160 + 1) + (1 if n >= N else 0) for n in lens)
    d = q.view(np.uint32) != w.view(np.uint32)
    assert not d.any(), f"{d.sum()} cells differ, columns {sorted(set(np.argwhere(d)[:, 1]))}, first rows {sorted(set(np.argwhere(d)[:, 0]))[:6]}"
