"""Parity of the HIP path (through the C ABI, libsmilehip.so) against
 (a) the committed golden vectors produced by the real reference binary and
 (b) the CPU oracle on the same seeded inputs.
Needs a real MI355X: run with `pytest -m gpu` through gpurun.
"""
import os

import numpy as np
import pytest

from tolerance import assert_parity, corpus_col_scale

pytestmark = pytest.mark.gpu

SYNTH_KEYS = ["u0_16000", "u1_16000", "u2_16000", "u3_16000", "u10_16000",
              "u7_399", "u7_400", "u7_401", "u7_559", "u7_560", "u7_561", "u7_1000"]


@pytest.fixture(scope="module")
def hip():
    from opensmile_amd import capi
    ctx = capi.Context(0)
    assert "gfx950" in ctx.name()
    return capi, ctx


@pytest.fixture(scope="module", params=["fast", "generic"])
def plan(hip, request):
    """Both kernels: the fast Nfft=512 kernel and the generic reference-order one."""
    capi, ctx = hip
    old = os.environ.get("SMILEHIP_FORCE_GENERIC")
    os.environ["SMILEHIP_FORCE_GENERIC"] = "1" if request.param == "generic" else "0"
    p = capi.Plan(ctx)
    if old is None:
        os.environ.pop("SMILEHIP_FORCE_GENERIC")
    else:
        os.environ["SMILEHIP_FORCE_GENERIC"] = old
    yield p
    p.close()


def test_geometry_bit_exact(plan):
    g = plan.geometry
    assert (g.frame_size, g.frame_step, g.fft_size, g.n_bins, g.n_static, g.n_out) == (400, 160, 512, 257, 13, 39)
    assert plan.num_frames(160000) == 998 and plan.num_frames(399) == 0 and plan.num_frames(560) == 2
    # frame time stamps: vIdx * 0.01 exactly as the reference computes them
    for t in (0, 1, 7, 997):
        assert plan.frame_time(t) == t * 0.01


def test_golden_batch_ragged(hip, plan, golden_synth):
    """All golden utterances (incl. empty / 1-frame / ragged ones) as ONE packed batch."""
    capi, _ = hip
    pcms = [golden_synth["pcm_" + k] for k in SYNTH_KEYS]
    off = np.concatenate([[0], np.cumsum([len(p) for p in pcms])]).astype(np.int64)
    b = capi.Batch(plan, off)
    refs = [golden_synth["out_" + k] for k in SYNTH_KEYS]
    # frame counts / row offsets: integer contract
    assert b.total_frames == sum(r.shape[0] for r in refs)
    np.testing.assert_array_equal(np.diff(b.frame_offsets), [r.shape[0] for r in refs])
    out = b.run_host(np.concatenate(pcms))
    s_col = corpus_col_scale(refs, 13)
    for i, k in enumerate(SYNTH_KEYS):
        o = out[b.frame_offsets[i]:b.frame_offsets[i + 1]]
        if refs[i].shape[0] == 0:
            assert o.shape[0] == 0
            continue
        assert_parity(o, refs[i], block=13, what=k, col_scale=s_col)
    b.close()


def test_vs_oracle_10s(hip, plan, oracle):
    """Config-2 utterance shape (10 s -> 998 frames) on seeded inputs vs the CPU oracle."""
    capi, _ = hip
    from opensmile_amd import synth
    us = [0, 1, 4, 10, 23]
    pcm, off = synth.corpus(1, 160000, first=us[0])
    pcms = [synth.utterance(u, 160000) for u in us]
    off = np.arange(len(us) + 1, dtype=np.int64) * 160000
    b = capi.Batch(plan, off)
    out = b.run_host(np.concatenate(pcms))
    cfg = oracle.default_cfg()
    worst = 0.0
    refs = [oracle.mfcc_chain(cfg, p) for p in pcms]
    s_col = corpus_col_scale(refs, 13)
    for i, u in enumerate(us):
        o = out[b.frame_offsets[i]:b.frame_offsets[i + 1]]
        e2, _ = assert_parity(o, refs[i], block=13, what=f"u{u}", col_scale=s_col)
        worst = max(worst, e2)
    print(f"worst per-frame-scaled error vs oracle: {worst:.3e}")
    b.close()


def test_delta_tail_bit_exact_given_static(hip, plan, oracle):
    """R13 is pure float arithmetic on the static block: given the GPU's own
    static columns the delta/accel columns must equal the oracle's tick-accurate
    chain bit for bit, for every length incl. the T<=3 quirk."""
    capi, _ = hip
    from opensmile_amd import synth
    lens = [400 + 160 * (T - 1) + 3 for T in (1, 2, 3, 4, 5, 8, 9, 10, 50)]
    pcms = [synth.utterance(30 + i, n) for i, n in enumerate(lens)]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    b = capi.Batch(plan, off)
    out = b.run_host(np.concatenate(pcms))
    for i in range(len(lens)):
        o = out[b.frame_offsets[i]:b.frame_offsets[i + 1]]
        dd = oracle.delta_chain(o[:, :13], 2, 2)
        assert np.array_equal(o[:, 13:26], dd[0]), f"delta T={o.shape[0]}"
        assert np.array_equal(o[:, 26:39], dd[1]), f"accel T={o.shape[0]}"
    b.close()


def test_fused_delta_equals_window_chain(hip, oracle):
    """The fast kernel with the two regression stages inside (lld_mfcc512<..., DELTA>, dword-aligned input) against (a) the same
    kernel followed by the window-chain kernel (SMILEHIP_NO_FUSED_DELTA=1 at batch creation): every cell the same bits, static
    coefficients included; (b) the oracle's tick-accurate chain on the GPU's own static columns. Every utterance length from 1 to 40
    frames (the short path's limit of 16, every residue mod 4), lengths around the tile cuts, one long utterance; then a batch
    large enough for several tiles per utterance."""
    capi, ctx = hip
    from opensmile_amd import synth
    plan = capi.Plan(ctx)
    Ts = list(range(1, 41)) + [63, 64, 65, 127, 128, 129, 250, 251, 252, 253, 997, 998, 1000, 1001, 3000]
    lens = [400 + 160 * (T - 1) + 2 * (i % 3) for i, T in enumerate(Ts)]          # even lengths: even offsets
    pcm = np.concatenate([synth.utterance(40 + i, n) for i, n in enumerate(lens)])
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    outs = {}
    for mode in ("fused", "chain"):
        if mode == "chain":
            os.environ["SMILEHIP_NO_FUSED_DELTA"] = "1"
        try:
            b = capi.Batch(plan, off)
        finally:
            os.environ.pop("SMILEHIP_NO_FUSED_DELTA", None)
        outs[mode] = b.run_host(pcm)
        fo = b.frame_offsets
        b.close()
    assert outs["fused"].shape == outs["chain"].shape and outs["fused"].shape[1] == 39
    d = outs["fused"].view(np.uint32) != outs["chain"].view(np.uint32)
    assert not d.any(), f"{d.sum()} cells differ, first at {np.argwhere(d)[:5]}"
    for i, T in enumerate(Ts):
        o = outs["fused"][fo[i]:fo[i + 1]]
        assert o.shape[0] == T
        dd = oracle.delta_chain(o[:, :13], 2, 2)
        assert np.array_equal(o[:, 13:26], dd[0]) and np.array_equal(o[:, 26:39], dd[1]), T
    # many equal utterances: the tile length comes out near a quarter of an utterance
    pcm2, off2 = synth.corpus_tiled(6000, 32000, n_unique=8)
    res = []
    for mode in ("fused", "chain"):
        if mode == "chain":
            os.environ["SMILEHIP_NO_FUSED_DELTA"] = "1"
        try:
            b = capi.Batch(plan, off2)
        finally:
            os.environ.pop("SMILEHIP_NO_FUSED_DELTA", None)
        res.append(b.run_host(pcm2))
        b.close()
    assert np.array_equal(res[0].view(np.uint32), res[1].view(np.uint32))
    plan.close()


def test_config1_44k_generic(hip, golden_config1, oracle):
    """Config 1 geometry (44.1 kHz: N=1103, H=441, Nfft=2048) runs on the generic
    kernel; input = the reference's example wav when oracle/_ref ships it,
    else a synthetic 44.1 kHz signal checked against the oracle."""
    capi, ctx = hip
    cfg = capi.mfcc12_0_d_a_config()
    cfg.sample_rate = 44100.0
    p = capi.Plan(ctx, cfg)
    g = p.geometry
    assert (g.frame_size, g.frame_step, g.fft_size) == (1103, 441, 2048)
    import wave
    wav = os.path.join(oracle.REF_DIR, "opensmile.wav")
    if os.path.exists(wav):
        with wave.open(wav, "rb") as w:
            pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").copy()
        ref = golden_config1["out"]
    else:
        from opensmile_amd import synth
        pcm = synth.utterance(3, 90112, fs=44100)
        oc = oracle.default_cfg()
        oc.sample_rate = 44100.0
        ref = oracle.mfcc_chain(oc, pcm)
    b = capi.Batch(p, np.array([0, len(pcm)], np.int64))
    out = b.run_host(pcm)
    assert out.shape == ref.shape
    assert_parity(out, ref, block=13, what="config1")
    b.close()
    p.close()


def test_linearity_free_properties_full_size(hip, plan, oracle):
    """Config-2 size (1000 x 10 s = 998 000 frames): size-independent properties.
    (1) batch-composition invariance: an utterance's rows do not depend on what
    else is in the batch or where it sits (every one of the 1000 utterances against its first occurrence; the 16 first
    occurrences against the oracle under the accuracy gate, so every value of the batch is covered); (2) shift-by-hop: frame t
    of x equals frame t-1 of x[hop:] bit for bit (static block)."""
    capi, _ = hip
    import torch
    from opensmile_amd import synth
    n_utt, S = 1000, 160000
    pcm, off = synth.corpus_tiled(n_utt, S, n_unique=16)
    b = capi.Batch(plan, off)
    assert b.total_frames == 998000
    d_pcm = torch.from_numpy(pcm).cuda()
    d_out = torch.empty((b.total_frames, 39), dtype=torch.float32, device="cuda")
    b.run_device(d_pcm.data_ptr(), d_out.data_ptr(), 39)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    assert np.isfinite(out).all()
    # (1) tiles repeat every 16 utterances: EVERY copy's rows are its first occurrence's bits ...
    per = out.reshape(n_utt, 998, 39)
    for u in range(16, n_utt):
        assert np.array_equal(per[u].view(np.uint32), per[u % 16].view(np.uint32)), u
    # ... and the 16 first occurrences pass the accuracy gate against the oracle: with (1), the values of all 998 000 frames
    if oracle is not None:
        cfg = oracle.default_cfg()
        refs = [oracle.mfcc_chain(cfg, pcm[u * S:(u + 1) * S]) for u in range(16)]
        s_col = corpus_col_scale(refs, 13)
        for u in range(16):
            assert_parity(per[u], refs[u], block=13, what=f"full size u{u}", col_scale=s_col)
    # and identical to the same utterance run alone
    b1 = capi.Batch(plan, np.array([0, S], np.int64))
    solo = b1.run_host(pcm[5 * S:6 * S])
    assert np.array_equal(solo, out[b.frame_offsets[5]:b.frame_offsets[6]])
    # (2) shift by one hop
    sh = capi.Batch(plan, np.array([0, S - 160], np.int64))
    shifted = sh.run_host(pcm[5 * S + 160:6 * S])
    assert np.array_equal(shifted[:, :13], solo[1:, :13])
    for x in (b, b1, sh):
        x.close()


def test_maximum_sizes(hip, plan, oracle):
    """Edge of the size range: (1) ONE 30-minute utterance (28.8 M samples, 179 998 frames,
    5 625 tiles of one utterance); (2) 150 000 one- and two-frame utterances in one batch
    (every tile is a partial tile, every utterance takes the tick-accurate short delta path)."""
    capi, _ = hip
    import torch
    from opensmile_amd import synth
    # (1)
    S = 28_800_000
    base = synth.utterance(2, 160000)
    pcm = np.tile(base, S // len(base))
    b = capi.Batch(plan, np.array([0, S], np.int64))
    T = (S - 400) // 160 + 1
    assert b.total_frames == T == 179998
    d_pcm = torch.from_numpy(pcm).cuda()
    d_out = torch.empty((T, 39), dtype=torch.float32, device="cuda")
    b.run_device(d_pcm.data_ptr(), d_out.data_ptr(), 39)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    assert np.isfinite(out).all()
    for t0 in (0, 1000 * 37, T - 98):                      # slices that start on a hop boundary
        sl = pcm[t0 * 160: t0 * 160 + 16000]
        ref = oracle.mfcc_chain(oracle.default_cfg(), sl)[:, :13]
        got = out[t0:t0 + ref.shape[0], :13]
        scale = np.abs(ref).max(axis=1, keepdims=True)
        assert (np.abs(got - ref) / scale).max() <= 1e-5
    # interior delta rows follow the closed form on the GPU's own static block
    t = 90000
    num = sum(i * (out[t + i, :13] - out[t - i, :13]) for i in (1, 2))
    assert np.allclose(out[t, 13:26], num / 10.0, rtol=0, atol=2e-6 * np.abs(out[t, :13]).max())
    b.close()
    del d_pcm, d_out
    # (2)
    n_utt = 150_000
    lens = np.where(np.arange(n_utt) % 2 == 0, 400, 560).astype(np.int64)
    off = np.concatenate([[0], np.cumsum(lens)])
    src = synth.utterance(3, 16000)
    pcm2 = np.concatenate([src[:400], src[100:660]] * (n_utt // 2))
    b = capi.Batch(plan, off)
    assert b.total_frames == n_utt // 2 * 3
    out = b.run_host(pcm2)
    r1 = oracle.mfcc_chain(oracle.default_cfg(), src[:400])
    r2 = oracle.mfcc_chain(oracle.default_cfg(), src[100:660])
    assert r1.shape == (1, 39) and r2.shape == (2, 39)
    fo = b.frame_offsets
    for u in (0, 1, 2, 77777, n_utt - 2, n_utt - 1):
        ref = r1 if u % 2 == 0 else r2
        got = out[fo[u]:fo[u + 1]]
        assert got.shape == ref.shape
        assert (np.abs(got - ref)).max() <= 1e-5 * np.abs(ref[:, :13]).max()
    # every even (odd) utterance is the same input: identical rows
    assert np.array_equal(out[fo[0]:fo[1]], out[fo[n_utt - 2]:fo[n_utt - 1]])
    assert np.array_equal(out[fo[1]:fo[2]], out[fo[n_utt - 1]:fo[n_utt]])
    b.close()


GEOMETRIES = [
    # frame, step, preemph, win (smilehip id), symmetric pad, use_power, first, last  -> which fast-kernel instantiation
    dict(fs=0.025, st=0.010, pe=1, win="ham", sym=0, pw=1, first=0, last=12),     # <13,7,pe,pw>   (MFCC12_0_D_A)
    dict(fs=0.032, st=0.010, pe=1, win="han", sym=0, pw=1, first=0, last=12),     # N=512: <16,8>
    dict(fs=0.020, st=0.010, pe=0, win="ham", sym=1, pw=1, first=1, last=14),     # symmetric zero pad, no pre-emphasis (ComParE front end)
    dict(fs=0.025, st=0.0125, pe=1, win="ham", sym=0, pw=0, first=1, last=12),    # H=200: <13,8>, magnitudes into the mel bank (IS09 front end)
    dict(fs=0.025, st=0.005, pe=1, win="gau", sym=1, pw=1, first=0, last=12),     # H=80, Gauss window, padded
    dict(fs=0.0301, st=0.010, pe=0, win="rec", sym=0, pw=0, first=0, last=5),     # odd N=482 -> (pad_left odd with sym=0 is 0) rectangular
]
WIN_IDS = {"rec": 0, "han": 1, "ham": 2, "gau": 3}    # same ids in smilehip.h and lld_oracle.h


@pytest.mark.parametrize("kind", ["fast", "generic"])
@pytest.mark.parametrize("geo", GEOMETRIES, ids=lambda g: f"{g['fs']}-{g['st']}-{g['win']}-pe{g['pe']}-sym{g['sym']}-pw{g['pw']}")
def test_other_geometries_and_options_vs_oracle(hip, oracle, geo, kind):
    """The template instantiations the BASELINE configs do not reach (N = 512, 8 PCM steps, no pre-emphasis, symmetric
    zero padding, magnitude mel input, other windows / hops), fast and reference-order kernels, against the oracle."""
    capi, ctx = hip
    from opensmile_amd import synth
    cfg = capi.mfcc12_0_d_a_config()
    oc = oracle.default_cfg()
    cfg.frame_size_sec = oc.frame_size_sec = geo["fs"]
    cfg.frame_step_sec = oc.frame_step_sec = geo["st"]
    cfg.preemph = oc.preemph_enable = geo["pe"]
    cfg.win_func = WIN_IDS[geo["win"]]
    oc.win_func = WIN_IDS[geo["win"]]
    cfg.win_sigma = oc.win_sigma = 0.4
    cfg.zero_pad_symmetric = oc.zero_pad_symmetric = geo["sym"]
    cfg.use_power = oc.use_power = geo["pw"]
    cfg.first_mfcc = oc.first_mfcc = geo["first"]
    cfg.last_mfcc = oc.last_mfcc = geo["last"]
    os.environ["SMILEHIP_FORCE_GENERIC"] = "1" if kind == "generic" else "0"
    try:
        plan = capi.Plan(ctx, cfg)
    finally:
        os.environ.pop("SMILEHIP_FORCE_GENERIC", None)
    D = geo["last"] - geo["first"] + 1
    lens = [16000, 7001, 3 * int(round(geo["fs"] * 16000))]
    pcms = [synth.utterance(60 + i, n) for i, n in enumerate(lens)]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    b = capi.Batch(plan, off)
    out = b.run_host(np.concatenate(pcms))
    for i, p in enumerate(pcms):
        ref = oracle.mfcc_chain(oc, p)
        got = out[b.frame_offsets[i]:b.frame_offsets[i + 1]]
        assert got.shape == ref.shape, (kind, geo, got.shape, ref.shape)
        scale = np.abs(ref[:, :D]).max(axis=1, keepdims=True)
        assert (np.abs(got - ref) / np.maximum(scale, 1e-30)).max() <= 1e-5, (kind, geo)
    b.close()
    plan.close()
