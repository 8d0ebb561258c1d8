"""BASELINE config 5 on the GPU: the eGeMAPSv02 chain (SMILEHIP_CHAIN_EGEMAPS, opensmile_amd/csrc/lld_gemaps.hip) against the
golden vectors of the real binary (tests/golden/egemaps_lld_synth.npz) and -- stage by stage, on identical inputs -- against
the CPU oracle (oracle/lld_oracle_gemaps.c).

How the tolerances are set. Everything the chain computes from the 20 ms spectrum directly (loudness, log-spectral
descriptors, flux, MFCC), the F0 contour and jitter / shimmer is well conditioned: gates at about twice the measured error
(profiles/r02_egemaps_parity.json: <= 2.2e-6 of the column scale). The formant columns are not: float32 Durbin on a 220-sample
frame followed by polynomial root finding amplifies the FFT's round-off (1e-7) to 1e-2 on ~1.5 % of the frames -- the CPU
oracle itself deviates from the binary by the same amount as soon as its FFT is not the reference's own
(test_oracle_pin_gemaps.py keeps that on record). For those stages parity is therefore shown per stage on IDENTICAL inputs
(bit-exact / 1e-6), and the chain-level gate bounds the share of frames that leave the 1e-5 band."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KEYS = ["u2_16000", "u3_48000", "u10_16000", "u1_16000", "u0_16000", "u7_960", "u7_1120", "u7_1600", "u7_2720", "u4_9000",
        "u37_9000", "u2_8720", "u5_16000", "u11_160000", "u3_1600", "u3_1760", "u10_1280", "u2_1760", "u28_2240",
        "u4_1920", "u10_1440", "u3_1280", "u7_800"]


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "egemaps_lld_synth.npz"))


@pytest.fixture(scope="module")
def gm():
    from opensmile_amd import capi
    ctx = capi.Context(0)
    plan = capi.Plan(ctx, capi.egemapsv02_config())
    yield capi, ctx, plan
    plan.close()
    ctx.close()


@pytest.fixture(scope="module")
def run_all(gm, golden):
    """All golden cases as one ragged batch (incl. one utterance without a 60 ms frame)."""
    capi, ctx, plan = gm
    pcms = [golden["pcm_" + k] for k in KEYS]
    off = np.concatenate([[0], np.cumsum([len(p) for p in pcms])]).astype(np.int64)
    b = capi.Batch(plan, off)
    lld, func, taps = b.run_host_egemaps(np.concatenate(pcms), taps=True)
    res = dict(batch=b, lld=lld, func=func, taps=taps, f20=b.frame_offsets_frames(), rows=b.frame_offsets.copy())
    yield res
    b.close()


def cat(golden, name, width):
    return np.concatenate([golden[name + "_" + k].reshape(-1, width) for k in KEYS if golden[name + "_" + k].size], axis=0)


def scaled_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return np.abs(a - b) / np.maximum(np.abs(b).max(axis=0), 1e-6)


def bits_equal(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return (a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))


# ------------------------------------------------------------------ integer contract
def test_row_counts_and_offsets(run_all, golden):
    """rows = T60 + 1 per utterance (0 without a 60 ms frame): frame indices are the bit-exact part of the contract."""
    rows = np.diff(run_all["rows"])
    for i, k in enumerate(KEYS):
        assert rows[i] == golden["lld_" + k].reshape(-1, 25).shape[0], k
    assert rows[KEYS.index("u7_800")] == 0
    assert run_all["lld"].shape == (int(rows.sum()), 25)


def test_pending_counts_match_the_oracle(run_all, golden, oracle):
    """P (frames the Viterbi pass had not decided at the end of input) drives every end-of-input rule of the graph."""
    for i, k in enumerate(KEYS):
        d = oracle.egemaps_levels(golden["pcm_" + k])
        if d["T60"] >= 1:
            assert run_all["taps"]["pending"][i] == d["P"], k


# ------------------------------------------------------------------ chain vs the real binary's levels
@pytest.mark.parametrize("level,cols,tol", [("loudness", slice(0, 1), 0.0), ("lspec", slice(1, 5), 0.0), ("flux", slice(5, 6), 0.0),
                                            ("mfcc", slice(6, 10), 0.0), ("energy2", slice(10, 11), 0.0)])
def test_20ms_levels_vs_golden(run_all, golden, level, cols, tol):
    """Round 3: the transform is the reference's rdft network, so every level that has no libm call downstream of it is
    the binary's bit for bit (tol 0 = bits_equal); so are the log-spectrum and log-mel levels since logf is glibc's algorithm
    (glibc_float.hpp) and cSpectral's FLOAT_DMEM sums are sequential float chains."""
    width = cols.stop - cols.start
    got, ref = run_all["taps"]["raw20"][:, cols], cat(golden, level, width)
    if tol == 0.0:
        assert bits_equal(got, ref).all(), f"{level}: {(~bits_equal(got, ref)).sum()} cells differ"
    else:
        e = scaled_err(got, ref)
        assert e.max() <= tol, f"{level}: {e.max():.3g}"


def test_pitch_level_vs_golden(run_all, golden):
    """gemapsv01b_logPitch [F0final, F0finalLog, voicing]: F0final and voicing are the binary's bits; F0finalLog goes
    through log(): <= 4e-7 of the column scale."""
    g, r = run_all["taps"]["pitch3"], cat(golden, "pitch", 3)
    assert bits_equal(g, r).all()


def test_jitter_shimmer_vs_golden(run_all, golden):
    t = run_all["taps"]
    jit = np.concatenate([t["jit4"][:, 0:1], t["shim_db"]], axis=1)
    assert bits_equal(jit, cat(golden, "jitter", 2)).all()


def test_lpc_formants_bit_identical_and_harmonics_vs_golden(run_all, golden):
    """cSpecResample -> cLpc -> cFormantLpc from the 20 ms spectrum: the binary's bits (round 2: 1.9 % of the formant cells
    beyond 1e-3 -- the root finder amplified the last bit of a different FFT order). Harmonics: dB values through log10,
    <= 1e-6 of the column scale."""
    t = run_all["taps"]
    assert bits_equal(t["lpc"][:, :11], cat(golden, "lpc", 11)).all()
    assert bits_equal(t["formants"], cat(golden, "formants", 10)).all()
    assert bits_equal(t["harm6"], cat(golden, "harm", 6)).all()      # dB values: log10f in glibc's order


def test_lld_level_vs_golden(run_all, golden):
    ref = np.concatenate([golden["lld_" + k].reshape(-1, 25) for k in KEYS], axis=0)
    same = bits_equal(run_all["lld"], ref)
    assert same.all(), f"{int((~same).sum())} of {same.size} cells differ from the binary, columns {sorted(set(np.argwhere(~same)[:, 1]))}"


# ------------------------------------------------------------------ stages on identical inputs
def test_stage_spectral_identical_input(gm, golden, oracle):
    capi, ctx, plan = gm
    _, t = oracle.mfcc_chain(oracle.frames_cfg(0.020, "ham"), golden["pcm_u3_48000"], taps=True)
    ref = oracle.egemaps_spectral_rows(t["mag"])
    out = capi.spectral_gemaps_host(plan, t["mag"])
    assert scaled_err(out, ref).max() <= 1e-6       # double tree sums instead of sequential ones


def test_stage_specresample_and_lpc_bit_exact(gm, golden, oracle):
    """Every float sum of smileDsp_irdft / smileDsp_autoCorr / Durbin in the reference's order: same bits as the oracle."""
    capi, ctx, plan = gm
    for key in ("u3_48000", "u10_16000", "u0_16000", "u1_16000"):
        _, t = oracle.mfcc_chain(oracle.frames_cfg(0.020, "ham"), golden["pcm_" + key], taps=True)
        x_ref = oracle.egemaps_specresample_rows(t["fft"])
        x = capi.specresample_host(plan, t["fft"])
        assert bits_equal(x, x_ref).all(), key
        assert bits_equal(capi.lpc_host(plan, x_ref), oracle.egemaps_lpc_rows(x_ref)).all(), key


def test_stage_formants_on_the_binarys_lpc(gm, golden):
    """The QR iteration in double, same operation sequence: the binary's own LP coefficients give the binary's formants."""
    capi, ctx, plan = gm
    lpc, ref = cat(golden, "lpc", 11), cat(golden, "formants", 10)
    out = capi.formantlpc_host(plan, lpc)
    assert bits_equal(out, ref).mean() >= 0.999
    assert scaled_err(out, ref).max() <= 1e-6


def test_stage_harmonics_identical_input(gm, golden, oracle):
    capi, ctx, plan = gm
    for key in ("u3_48000", "u2_16000", "u4_9000"):
        _, t60 = oracle.mfcc_chain(oracle.frames_cfg(0.060, "gauss"), golden["pcm_" + key], taps=True)
        f0 = golden["pitch_" + key][:, 0]
        fm = golden["formants_" + key][:len(f0)]
        ref = oracle.egemaps_harmonics_rows(f0, fm, t60["mag"])
        out = capi.harmonics_host(plan, f0, fm, t60["mag"])
        e = scaled_err(out, ref)
        # the harmonic search and the HNR's ACF (the reference's inverse rdft network) are exact on identical magnitudes
        assert e.max() <= 1e-6, key


# ------------------------------------------------------------------ selectors / smoothers / functionals
def test_tail_equals_oracle_on_the_devices_levels(run_all, golden, oracle):
    """Selectors, gates and the nine smoothers are pure data movement + 3-point means: fed with the device's own per-frame
    levels the oracle's smoothing (with the end-of-input rules pinned against the binary) must give the device's rows."""
    import ctypes as C
    t, f20, f60 = run_all["taps"], run_all["f20"], run_all["taps"]["frame_off60"]
    L = oracle.lib()
    oracle._eg_bind()
    for i, k in enumerate(KEYS):
        T20, T60 = int(f20[i + 1] - f20[i]), int(f60[i + 1] - f60[i])
        if T60 < 1:
            continue
        raw = t["raw20"][f20[i]:f20[i + 1]]
        lv = oracle._EgLv()
        keep = {}

        def put(name, a):
            a = np.ascontiguousarray(a, np.float32)
            keep[name] = a
            setattr(lv, name, a.ctypes.data_as(C.POINTER(C.c_float)))
        lv.T20, lv.T60, lv.P = T20, T60, int(t["pending"][i])
        put("loudness", raw[:, 0]); put("lspec", raw[:, 1:5]); put("flux", raw[:, 5]); put("mfcc", raw[:, 6:10])
        put("energy2", raw[:, 10]); put("formants", t["formants"][f20[i]:f20[i + 1]])
        put("pitch", t["pitch3"][f60[i]:f60[i + 1]])
        put("jitter", np.concatenate([t["jit4"][f60[i]:f60[i + 1], 0:1], t["shim_db"][f60[i]:f60[i + 1]]], axis=1))
        put("harm", t["harm6"][f60[i]:f60[i + 1]])
        sm = oracle._EgSmo()
        L.lldo_egemaps_smooth(C.byref(lv), C.byref(sm))
        E = oracle._eg_arr(sm.E, T20 + 1, 10); F = oracle._eg_arr(sm.F, T60 + 1, 15)
        ref = np.concatenate([E[:T60 + 1], F], axis=1)
        out = run_all["lld"][run_all["rows"][i]:run_all["rows"][i + 1]]
        assert bits_equal(out, ref).all(), k
        fin = t["func_in"][t["fin_off"][i]:t["fin_off"][i + 1]]
        assert bits_equal(fin[:, 0], E[:, 0]).all() and bits_equal(fin[:, 1:6], E[:, 5:10]).all(), k
        assert bits_equal(fin[:T60 + 1, 6:7], oracle._eg_arr(sm.logf0, T60 + 1, 1)).all(), k
        assert bits_equal(fin[:T60 + 1, 7:21], oracle._eg_arr(sm.NoNz, T60 + 1, 14)).all(), k
        assert bits_equal(fin[:T60 + 1, 21:30], oracle._eg_arr(sm.specV, T60 + 1, 9)).all(), k
        assert bits_equal(fin[:T60 + 1, 30:35], oracle._eg_arr(sm.specU, T60 + 1, 5)).all(), k
        # the functionals on those levels: device == oracle (libm-dependent values to 1e-6)
        fo = np.zeros(88, np.float32)
        assert L.lldo_egemaps_func_from_levels(C.byref(lv), C.byref(sm), fo.ctypes.data) == 1
        rel = np.abs(run_all["func"][i] - fo) / np.maximum(np.abs(fo), 1e-3)
        assert rel.max() <= 2e-6, (k, int(np.argmax(rel)), rel.max())
        L.lldo_egemaps_smo_free(C.byref(sm))


def test_functionals_vs_golden(run_all, golden):
    """The 88 functionals against the real binary: utterances without a 60 ms frame give zeros (no vector in the
    reference); otherwise the well-conditioned values agree to 1e-4, the formant-dependent ones statistically."""
    fr, fg = [], []
    for i, k in enumerate(KEYS):
        r = golden["func_" + k].reshape(-1, 88)
        if r.shape[0] == 0:
            assert not run_all["func"][i].any(), k
        else:
            fr.append(r[0]); fg.append(run_all["func"][i])
    fr, fg = np.array(fr), np.array(fg)
    rel = np.abs(fg - fr) / np.maximum(np.abs(fr), 1e-2)
    # F0 (0..9), loudness (10..19), flux / mfcc mean + stddevNorm (20..29), temporal set (81..86), leq (87)
    assert bits_equal(fg, fr).all(), (rel.max(), int(np.argmax(rel.max(axis=0))))      # round 2: 93 % within 1e-3


# ------------------------------------------------------------------ batch properties at a larger size
def test_large_batch_is_deterministic_and_order_independent(gm):
    """300 x 3 s (one thousandth of config 5's per-GPU share): two runs give identical bits, and an utterance's rows do not
    depend on its neighbours in the batch."""
    from opensmile_amd import synth
    capi, ctx, plan = gm
    n = 300
    pcms = [synth.utterance(100 + (u % 40), 48000) for u in range(n)]
    off = np.concatenate([[0], np.cumsum([len(p) for p in pcms])]).astype(np.int64)
    b = capi.Batch(plan, off)
    pcm = np.concatenate(pcms)
    lld1, f1 = b.run_host_egemaps(pcm)
    lld2, f2 = b.run_host_egemaps(pcm)
    assert np.array_equal(lld1.view(np.uint32), lld2.view(np.uint32)) and np.array_equal(f1.view(np.uint32), f2.view(np.uint32))
    r = b.frame_offsets
    assert np.all(np.diff(r) == 296)
    for u in (0, 7, 123):                           # same waveform as utterance u + 40
        assert np.array_equal(lld1[r[u]:r[u + 1]], lld1[r[u + 40]:r[u + 41]])
        assert np.array_equal(f1[u], f1[u + 40])
    assert np.isfinite(lld1).all() and np.isfinite(f1).all()
    b.close()
