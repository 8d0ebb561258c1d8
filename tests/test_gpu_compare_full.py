"""Config-4 shape: the whole LLD level of ComParE_2016 (130 columns: F0 group incl. cPitchJitter, groups A and B, and
their deltas; T60+1 rows) through the C ABI's smilehip_lld_run (chain_kind COMPARE), against golden outputs of the
real reference binary and against the CPU oracle."""
import numpy as np
import pytest

from test_oracle_pin_compare import compare_tolerances
from test_oracle_pin_f0 import KEYS_130

pytestmark = pytest.mark.gpu

AB = list(range(6, 65)) + list(range(71, 130))          # groups A+B [sma | delta]
F0 = list(range(0, 6)) + list(range(65, 71))


@pytest.fixture(scope="module")
def hip():
    from opensmile_amd import capi
    ctx = capi.Context(0)
    plan = capi.Plan(ctx, capi.compare16_config())
    assert (plan.geometry.n_static, plan.geometry.n_out) == (65, 130)
    return capi, ctx, plan


def f0_lld_tolerances(out, ref, what=""):
    """The 12 F0-group columns (F0final, voicing, jitterLocal, jitterDDP, shimmerLocal, logHNR and their deltas): the reference's
    bits (profiles/r03_compare_parity.json: 32 770 rows of fresh utterances against the real binary, every cell identical).
    Round 2's gate was statistical: <= 1 % of the rows beyond 1e-5, nothing beyond 2e-4."""
    from tolerance import assert_bits_equal
    assert_bits_equal(out, ref, what)


def test_compare_full_golden_batch_ragged(hip, golden_f0):
    capi, ctx, plan = hip
    pcms = [golden_f0["pcm_" + k] for k in KEYS_130]
    refs = [golden_f0["lld130_" + k] for k in KEYS_130]
    off = np.concatenate([[0], np.cumsum([len(p) for p in pcms])]).astype(np.int64)
    b = capi.Batch(plan, off)
    np.testing.assert_array_equal(np.diff(b.frame_offsets), [r.shape[0] for r in refs])
    out = b.run_host(np.concatenate(pcms))
    for i, k in enumerate(KEYS_130):
        o = out[b.frame_offsets[i]:b.frame_offsets[i + 1]]
        compare_tolerances(o[:, AB], refs[i][:, AB], k)
        f0_lld_tolerances(o[:, F0], refs[i][:, F0], k)
    b.close()


def test_compare_full_vs_oracle_ragged_lengths(hip, oracle):
    capi, ctx, plan = hip
    from opensmile_amd import synth
    lens = [160000, 100, 959, 960, 1439, 1440, 1600, 1760, 2720, 9000, 8720, 21280, 160000, 48000, 0]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pcm = np.concatenate([synth.utterance(50 + i, n) if n else np.zeros(0, np.int16) for i, n in enumerate(lens)])
    b = capi.Batch(plan, off)
    out = b.run_host(pcm)
    assert out.shape[1] == 130
    oracle.use_reference_fft(False)
    for i, n in enumerate(lens):
        ref = oracle.compare_lld_chain(pcm[off[i]:off[i + 1]])
        o = out[b.frame_offsets[i]:b.frame_offsets[i + 1]]
        assert o.shape == ref.shape, (n, o.shape, ref.shape)
        if ref.shape[0]:
            compare_tolerances(o[:, AB], ref[:, AB], f"len{n}")
            f0_lld_tolerances(o[:, F0], ref[:, F0], f"len{n}")
    b.close()


def oracle_f0_lld_from_levels(x, P):
    """lldo_compare_f0_lld's smoothing / delta rules (oracle/lld_oracle_f0.c) on given levels: x = T x 6."""
    f32 = np.float32
    T = x.shape[0]
    rows = T + 1
    z = np.zeros((rows, 12), f32)
    for n in range(rows):
        for d in range(6):
            clip = T - 1
            if d >= 2 and n <= T - P and P < T:
                clip = T - P - 1
            clip = max(clip, 0)
            X = lambda i: x[min(max(i, 0), clip), d]       # noqa: E731
            c = X(n)
            if c != 0:
                s, cnt = f32(c), 1
                if X(n - 1) != 0:
                    s = f32(s + X(n - 1))
                    cnt += 1
                if X(n + 1) != 0:
                    s = f32(s + X(n + 1))
                    cnt += 1
                z[n, d] = f32(s / f32(cnt))
    norm = f32(10.0)
    for n in range(rows):
        cl = T if P >= T else ((T - P) if n <= T - P + 2 else (T - 1 if n == T - P + 3 else T))
        cl = min(max(cl, 0), rows - 1)
        for d in range(6):
            num = f32(0)
            for i in (1, 2):
                a = z[min(max(n - i, 0), cl), d]
                bb = z[min(n + i, cl), d]
                if a != 0 and bb != 0:
                    num = f32(num + f32(i) * f32(bb - a))
                    norm = f32(norm + f32(i * i))
            z[n, 6 + d] = f32(num / norm)
    return z


def test_jitter_and_lld_columns_exact_on_identical_f0(hip, oracle):
    """cPitchJitter, the noZeroSma smoothing and the onlyInSegments deltas are discrete / order-sensitive: fed with
    the SAME F0 contour and candidate rows (taken from the F0 chain run on the same input), device and oracle must
    agree bit for bit on the F0 group's 12 columns."""
    capi, ctx, plan = hip
    from opensmile_amd import synth
    import ctypes as C
    lens = [48000, 9000, 24000]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pcm = np.concatenate([synth.utterance(30 + i, n) for i, n in enumerate(lens)])
    b = capi.Batch(plan, off)
    out = b.run_host(pcm)
    plan_f0 = capi.Plan(ctx, capi.compare16_f0_config())
    bf = capi.Batch(plan_f0, off)
    pitch, taps = bf.f0_run_host_taps(pcm)
    L = oracle.lib()
    L.lldo_pitch_viterbi.restype = None
    L.lldo_pitch_viterbi.argtypes = [C.c_void_p, C.c_long, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    for i in range(len(lens)):
        sl = slice(bf.frame_offsets[i], bf.frame_offsets[i + 1])
        p2 = np.ascontiguousarray(pitch[sl])
        T = p2.shape[0]
        jit = oracle.pitch_jitter(pcm[off[i]:off[i + 1]], p2[:, 0])
        shs = np.ascontiguousarray(taps["shs"][sl])
        P = C.c_long(0)
        tmp = np.zeros((T, 2), np.float32)
        L.lldo_pitch_viterbi(shs.ctypes.data, T, np.float32(0.7), tmp.ctypes.data, None, C.byref(P))
        ref = oracle_f0_lld_from_levels(np.concatenate([p2, jit], axis=1), P.value)
        o = np.ascontiguousarray(out[b.frame_offsets[i]:b.frame_offsets[i + 1]][:, F0])
        assert o.shape == ref.shape
        assert np.array_equal(o.view(np.uint32), ref.view(np.uint32)), f"utt {i}: rows {sorted(set(np.argwhere(o != ref)[:, 0]))[:8]}"
    b.close()
    bf.close()


def test_jitter_forms_agree_bit_for_bit(hip, monkeypatch):
    """cPitchJitter's three forms -- runs of voiced frames taken from a counter (the default), one workgroup per utterance
    (SMILEHIP_JITTER=utt, the form the stream mode and the redo pass use), and the runs with every utterance handed to the
    redo pass (SMILEHIP_JITTER=redo) -- write the same bits, on ragged lengths incl. utterances without a 60 ms frame, a
    noise-only one (no voiced frame), the all-zero one and the square wave (one run over the whole utterance); a second
    run of the default form checks that the counters were left at zero."""
    capi, ctx, plan = hip
    from opensmile_amd import synth
    lens = [160000, 100, 959, 48000, 1760, 160000, 9000, 160000, 0, 21280, 160000]
    seeds = [0, 3, 4, 5, 6, 1, 7, 10, 8, 9, 12]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pcm = np.concatenate([synth.utterance(s, n) if n else np.zeros(0, np.int16) for s, n in zip(seeds, lens)])
    b = capi.Batch(plan, off)
    monkeypatch.delenv("SMILEHIP_JITTER", raising=False)
    base = b.run_host(pcm)[:, F0].copy()
    assert np.isfinite(base).all() and (base[:, 2:6] != 0).any()
    for mode in ("utt", "redo", None, None):
        if mode:
            monkeypatch.setenv("SMILEHIP_JITTER", mode)
        else:
            monkeypatch.delenv("SMILEHIP_JITTER", raising=False)
        o = b.run_host(pcm)[:, F0]
        assert np.array_equal(o.view(np.uint32), base.view(np.uint32)), f"mode {mode}: rows {sorted(set(np.argwhere(o != base)[:, 0]))[:8]}"
    b.close()


def test_compare_full_large_batch_properties(hip, oracle):
    """A per-GPU share in the direction of config 4 (3000 x 10 s = 3.0 M rows x 130 columns): size-independent
    properties -- row counts, copies of the same utterance give bit-identical rows wherever they sit in the batch,
    value ranges of the F0 group -- and one utterance checked against the oracle."""
    capi, ctx, plan = hip
    from opensmile_amd import synth
    n_utt, S, n_unique = 3000, 160000, 12
    pcm, off = synth.corpus_tiled(n_utt, S, n_unique=n_unique)
    b = capi.Batch(plan, off)
    assert b.total_rows == n_utt * 996
    out = b.run_host(pcm).reshape(n_utt, 996, 130)
    assert np.isfinite(out).all()
    for u in range(n_unique):
        same = (out[u::n_unique].view(np.uint32) == out[u].view(np.uint32)[None]).all()
        assert same, f"copies of utterance {u} differ"
    f0, voi, jl, jd, sh, hnr = (out[..., c] for c in range(6))
    assert ((f0 == 0) | ((f0 >= 52.0 * 0.999) & (f0 <= 620.0 * 1.001))).all()
    assert (voi >= 0).all() and (voi <= 1.0).all()
    for x in (jl, jd, sh):
        assert (x >= 0).all() and (x <= 1.0).all()
    assert (hnr >= -100.0).all()
    oracle.use_reference_fft(False)
    ref = oracle.compare_lld_chain(pcm[off[5]:off[6]])
    compare_tolerances(out[5][:, AB], ref[:, AB], "large batch, utterance 5")
    f0_lld_tolerances(out[5][:, F0], ref[:, F0], "large batch, utterance 5")
    b.close()
