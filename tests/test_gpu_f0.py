"""the F0 group of ComParE_2016 (cSpecScale -> cPitchShs -> cPitchSmootherViterbi ->
cValbasedSelector on the 60 ms frames) through the C ABI's smilehip_lld_run (chain_kind COMPARE_F0), against
golden levels of the real reference binary and against the CPU oracle, level by level."""
import numpy as np
import pytest

from test_oracle_pin_f0 import KEYS, f0_tolerances

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from opensmile_amd import capi
    ctx = capi.Context(0)
    plan = capi.Plan(ctx, capi.compare16_f0_config())
    g = plan.geometry
    assert (g.frame_size, g.frame_step, g.fft_size, g.n_bins, g.n_out) == (960, 160, 1024, 513, 2)
    return capi, ctx, plan


def level_tolerances(taps, ref_hps, ref_shs, ref_e60, what):
    """The internal levels of the F0 group -- cSpecScale's output (hps), cPitchShs's candidates (shs), the 60 ms frame energy --
    against the real binary's: identical bits (round 2: hps within 1e-5, <= 1 % of the candidate rows allowed to differ)."""
    from tolerance import assert_bits_equal
    if ref_hps is not None and ref_hps.size:
        assert_bits_equal(taps["hps"], ref_hps, f"{what}: hps")
    if ref_e60.size:
        assert_bits_equal(taps["e60"].reshape(ref_e60.shape), ref_e60, f"{what}: e60")
    if ref_shs.size:
        assert_bits_equal(taps["shs"], ref_shs, f"{what}: shs")


def test_f0_golden_batch_ragged(hip, golden_f0):
    capi, ctx, plan = hip
    pcms = [golden_f0["pcm_" + k] for k in KEYS]
    off = np.concatenate([[0], np.cumsum([len(p) for p in pcms])]).astype(np.int64)
    b = capi.Batch(plan, off)
    np.testing.assert_array_equal(np.diff(b.frame_offsets), [golden_f0["pitch_" + k].shape[0] for k in KEYS])   # T60 rows
    out, taps = b.f0_run_host_taps(np.concatenate(pcms))
    for i, k in enumerate(KEYS):
        sl = slice(b.frame_offsets[i], b.frame_offsets[i + 1])
        t = {n: v[sl] for n, v in taps.items()}
        hps = golden_f0["hps_" + k] if "hps_" + k in golden_f0.files else None
        level_tolerances(t, hps, golden_f0["shs_" + k], golden_f0["e60_" + k], k)
        f0_tolerances(out[sl], golden_f0["pitch_" + k], k)
    b.close()


def test_f0_vs_oracle_ragged_lengths(hip, oracle):
    """10 s utterances, inputs shorter than one 60 ms frame, one/two/three frames, lengths around the 30-frame
    Viterbi buffer (forced decisions start at 31 frames)."""
    capi, ctx, plan = hip
    from opensmile_amd import synth
    lens = [160000, 100, 959, 960, 1120, 1280, 5600, 5760, 5920, 9600, 160000, 48000, 0, 80000]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pcm = np.concatenate([synth.utterance(70 + i, n) if n else np.zeros(0, np.int16) for i, n in enumerate(lens)])
    b = capi.Batch(plan, off)
    out, taps = b.f0_run_host_taps(pcm)
    assert out.shape[1] == 2
    oracle.use_reference_fft(False)
    for i, n in enumerate(lens):
        ref, rt = oracle.compare_f0_chain(pcm[off[i]:off[i + 1]], taps=True)
        sl = slice(b.frame_offsets[i], b.frame_offsets[i + 1])
        assert out[sl].shape == ref.shape, (n, out[sl].shape, ref.shape)
        if ref.shape[0]:
            level_tolerances({k: v[sl] for k, v in taps.items()}, rt["hps"], rt["shs"], rt["e60"], f"len{n}")
            f0_tolerances(out[sl], ref, f"len{n}")
    b.close()


def test_f0_viterbi_exact_on_identical_candidates(hip, oracle):
    """The Viterbi pass and the energy gate are discrete: fed with the SAME candidate rows (the device's own
    is13_pitchShsG60 level), device and oracle must pick the same path and emit identical values."""
    capi, ctx, plan = hip
    from opensmile_amd import synth
    lens = [160000, 4800, 5920, 16000, 64000]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pcm = np.concatenate([synth.utterance(90 + i, n) for i, n in enumerate(lens)])
    b = capi.Batch(plan, off)
    out, taps = b.f0_run_host_taps(pcm)
    import ctypes as C
    L = oracle.lib()
    L.lldo_pitch_viterbi.restype = None
    L.lldo_pitch_viterbi.argtypes = [C.c_void_p, C.c_long, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    for i in range(len(lens)):
        sl = slice(b.frame_offsets[i], b.frame_offsets[i + 1])
        shs = np.ascontiguousarray(taps["shs"][sl])
        T = shs.shape[0]
        ref = np.zeros((T, 2), np.float32)
        L.lldo_pitch_viterbi(shs.ctypes.data, T, np.float32(0.7), ref.ctypes.data, None, None)
        ref[~(taps["e60"][sl, 0] > np.float32(0.001))] = 0.0
        assert np.array_equal(out[sl].view(np.uint32), ref.view(np.uint32)), f"utt {i}: {(out[sl] != ref).any(axis=1).sum()} rows"
    b.close()


def test_f0_rerun_is_deterministic(hip):
    capi, ctx, plan = hip
    from opensmile_amd import synth
    pcm = np.concatenate([synth.utterance(3, 32000), synth.utterance(4, 16000)])
    b = capi.Batch(plan, np.array([0, 32000, 48000], np.int64))
    a = b.run_host(pcm)
    c = b.run_host(pcm)
    assert np.array_equal(a.view(np.uint32), c.view(np.uint32))
    b.close()


def test_viterbi_stream_equals_the_oracle_frame_by_frame(hip, oracle):
    """smilehip_viterbi_stream_*: one frame per call, the trellis on the device between the calls. Every decided frame
    carries the state the oracle's pass picks (identical F0 / voicing), and the frames become decided at exactly the pushes
    at which the incremental reference algorithm releases them (the oracle's `pending` count at the end)."""
    capi, ctx, plan = hip
    from opensmile_amd import synth
    import ctypes as C
    L = oracle.lib()
    L.lldo_pitch_viterbi_ex.restype = None
    L.lldo_pitch_viterbi_ex.argtypes = [C.c_void_p, C.c_long, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    for buflen, n_samp in ((30, 64000), (40, 48000), (30, 4800), (90, 64000), (128, 96000)):      # > 64: several rounds of the decision scan
        pcm = synth.utterance(70 + buflen, n_samp)
        b = capi.Batch(plan, np.array([0, n_samp], np.int64))
        _, taps = b.f0_run_host_taps(pcm)
        shs = np.ascontiguousarray(taps["shs"])
        b.close()
        T = shs.shape[0]
        ref = np.zeros((T, 2), np.float32)
        states = np.zeros(T, np.int32)
        pend = C.c_long(0)
        L.lldo_pitch_viterbi_ex(shs.ctypes.data, T, np.float32(0.7), buflen, ref.ctypes.data, states.ctypes.data, C.byref(pend))
        vs = capi.ViterbiStream(ctx, buffer_len=buflen)
        got = {}
        before_flush = 0
        for t in range(T):
            for n, s in vs.push(shs[t, 1:7], shs[t, 7:13]):
                assert n not in got
                got[n] = s
            before_flush = len(got)
        for n, s in vs.flush():
            assert n not in got
            got[n] = s
        vs.close()
        assert sorted(got) == list(range(T))
        assert T - before_flush == pend.value                         # what only the flush at end of input decides
        f0 = np.array([shs[n, 1 + got[n]] if got[n] < 6 else 0.0 for n in range(T)], np.float32)
        vp = np.array([shs[n, 7 + got[n]] if got[n] < 6 else shs[n, 7] for n in range(T)], np.float32)
        assert np.array_equal(f0.view(np.uint32), ref[:, 0].view(np.uint32)) and np.array_equal(vp.view(np.uint32), ref[:, 1].view(np.uint32))


def test_f0_exact_tree_mean_equals_the_sequential_chain(hip):
    """lld_f0_cand forms the mean of the summation spectrum as a tree sum when it can prove the sum exact (non-negative floats
    whose smallest non-zero value's last bit keeps every partial sum representable) and runs the reference's sequential chain
    otherwise. The per-component operator smilehip_pitchshs_frames always runs the sequential chain: fed with the chain's own
    octave-scale spectra it must give the chain's 21 values bit for bit -- on an all-zero utterance (S = 0), a voiced one and a
    noise one (whichever path the chain's test sends each frame down; the fallback is the operator's own code)."""
    import torch
    capi, ctx, plan = hip
    from opensmile_amd import synth
    L = capi.load()
    lens = [160000, 32000, 16000]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pcm = np.concatenate([synth.utterance(0, lens[0]), synth.utterance(7, lens[1]), synth.utterance(10, lens[2])])   # zeros | voiced | noise
    b = capi.Batch(plan, off)
    _, taps = b.f0_run_host_taps(pcm)
    b.close()
    hps, shs = taps["hps"], taps["shs"]
    nH, K = hps.shape
    d_hp = torch.from_numpy(np.ascontiguousarray(hps)).cuda()
    d_s = torch.zeros((nH, 21), dtype=torch.float32, device="cuda")
    assert L.smilehip_pitchshs_frames(plan._h, d_hp.data_ptr(), K, d_s.data_ptr(), 21, nH, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d_s.cpu().numpy().view(np.uint32), shs.view(np.uint32))
