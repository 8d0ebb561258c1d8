"""Stage-level parity through the C ABI's per-component entry points (the ones
the openSMILE plugin binds). Inputs are the ORACLE's intermediate levels, so
each HIP stage is checked in isolation: every stage whose arithmetic order is
the reference's own must be BIT-EXACT; the FFT (own butterfly order, FMA) is
checked against the oracle's FFT and a float64 DFT."""
import numpy as np
import pytest
import torch

from tolerance import RTOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from opensmile_amd import capi, synth
    from oracle import lldo
    ctx = capi.Context(0)
    plan = capi.Plan(ctx)
    cfg = lldo.default_cfg()
    lldo.use_reference_fft(False)
    pcm = np.concatenate([synth.utterance(u, 16000) for u in (2, 10, 1)])
    out, taps = lldo.mfcc_chain(cfg, pcm, taps=True)
    return capi, ctx, plan, lldo, cfg, pcm, out, taps


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_r0_pcm16_exhaustive_bit_exact(env):
    capi, ctx = env[0], env[1]
    s = np.arange(-32768, 32768, dtype=np.int16)
    d_in, d_out = dev(s), torch.empty(65536, dtype=torch.float32, device="cuda")
    capi.pcm16_to_float(ctx, d_in.data_ptr(), 65536, d_out.data_ptr())
    torch.cuda.synchronize()
    ref = s.astype(np.float32) / np.float32(32767.0)
    assert np.array_equal(bits(d_out.cpu().numpy()), bits(ref))


def test_r2_r3_preemphasis_window_bit_exact(env):
    capi, ctx, plan, lldo, cfg, pcm, out, taps = env
    T = taps["win"].shape[0]
    x = (pcm.astype(np.float32) / np.float32(32767.0))
    frames = np.stack([x[t * 160:t * 160 + 400] for t in range(T)])
    d_fr = dev(frames)
    d_pe = torch.empty_like(d_fr)
    d_w = torch.empty_like(d_fr)
    capi.preemphasis_frames(ctx, d_fr.data_ptr(), 400, d_pe.data_ptr(), 400, T, 400, float(np.float32(0.97)))
    capi.window_frames(plan, d_pe.data_ptr(), 400, d_w.data_ptr(), 400, T)
    torch.cuda.synchronize()
    assert np.array_equal(bits(d_w.cpu().numpy()), bits(taps["win"]))


def test_r4_rfft_vs_oracle_and_float64(env):
    capi, ctx, plan, lldo, cfg, pcm, out, taps = env
    T = taps["win"].shape[0]
    d_w = dev(taps["win"])
    d_f = torch.empty((T, 512), dtype=torch.float32, device="cuda")
    capi.rfft_frames(plan, d_w.data_ptr(), 400, d_f.data_ptr(), 512, T)
    torch.cuda.synchronize()
    got = d_f.cpu().numpy()
    # Ooura packing (fftsg.c:103-135): a[0]=Re X0, a[1]=Re X[256], a[2k]=Re Xk, a[2k+1]=-Im Xk
    X = np.fft.rfft(np.pad(taps["win"].astype(np.float64), ((0, 0), (0, 112))), axis=1)
    ref64 = np.zeros((T, 512))
    ref64[:, 0] = X[:, 0].real
    ref64[:, 1] = X[:, 256].real
    ref64[:, 2::2] = X[:, 1:256].real
    ref64[:, 3::2] = -X[:, 1:256].imag
    scale = np.abs(ref64).max(axis=1, keepdims=True)
    nz = scale[:, 0] > 0
    err_gpu = (np.abs(got - ref64)[nz] / scale[nz]).max()
    err_orc = (np.abs(taps["fft"] - ref64)[nz] / scale[nz]).max()
    print(f"rfft error vs float64 DFT, scaled by the frame's largest bin: HIP {err_gpu:.2e}, oracle FFT {err_orc:.2e}")
    assert err_gpu < 1e-6
    assert np.array_equal(got[~nz], taps["fft"][~nz])      # all-zero frames stay exactly zero


def test_r5_r6_r7_bit_exact_from_oracle_spectrum(env):
    capi, ctx, plan, lldo, cfg, pcm, out, taps = env
    T = taps["fft"].shape[0]
    d_f = dev(taps["fft"])
    d_m = torch.empty((T, 257), dtype=torch.float32, device="cuda")
    d_b = torch.empty((T, 26), dtype=torch.float32, device="cuda")
    d_c = torch.empty((T, 13), dtype=torch.float32, device="cuda")
    capi.fftmag_frames(plan, d_f.data_ptr(), 512, d_m.data_ptr(), 257, T)
    capi.melspec_frames(plan, d_m.data_ptr(), 257, d_b.data_ptr(), 26, T)
    capi.mfcc_frames(plan, d_b.data_ptr(), 26, d_c.data_ptr(), 13, T)
    torch.cuda.synchronize()
    assert np.array_equal(bits(d_m.cpu().numpy()), bits(taps["mag"])), "R5 magnitude"
    assert np.array_equal(bits(d_b.cpu().numpy()), bits(taps["mel"])), "R6 mel bank"
    # R7: the DCT/lifter arithmetic is the reference's, but its log is libm's logf
    # (glibc: <= 0.82 ulp, not always correctly rounded) while the kernel rounds the
    # double-precision log once (correctly rounded): a few log-mel inputs differ by
    # 1 ulp, i.e. ~1e-6 absolute on a cepstral coefficient.
    got, ref = d_c.cpu().numpy(), out[:, :13]
    same = (bits(got) == bits(ref)).mean()
    scale = np.abs(ref).max(axis=1, keepdims=True)
    err = (np.abs(got.astype(np.float64) - ref) / np.maximum(scale, 1e-30)).max()
    print(f"R7: {same * 100:.2f}% of coefficients bit-identical, worst per-frame-scaled error {err:.2e}")
    assert same > 0.9 and err < 1e-6


@pytest.mark.parametrize("W,orders", [(1, 1), (2, 2), (3, 2), (2, 1), (4, 2)])
def test_r13_delta_chain_bit_exact(env, W, orders):
    capi, ctx, plan, lldo = env[0], env[1], env[2], env[3]
    rng = np.random.default_rng(W * 10 + orders)
    lens = [1, 2, 3, 4, 5, 4 * W, 4 * W + 1, 17, 300, 129, 128, 127]
    D = 13
    off = np.concatenate([[0], np.cumsum([400 + 160 * (T - 1) for T in lens])]).astype(np.int64)
    b = capi.Batch(plan, off)
    assert list(np.diff(b.frame_offsets)) == lens
    x = rng.normal(size=(sum(lens), D)).astype(np.float32)
    io = np.zeros((sum(lens), D * (1 + orders)), np.float32)
    io[:, :D] = x
    d_io = dev(io)
    capi.delta_chain(plan, b, d_io.data_ptr(), D * (1 + orders), D, W, orders)
    torch.cuda.synchronize()
    got = d_io.cpu().numpy()
    for i, T in enumerate(lens):
        r0 = b.frame_offsets[i]
        ref = lldo.delta_chain(x[r0:r0 + T], W, orders)
        for o in range(orders):
            assert np.array_equal(bits(got[r0:r0 + T, D * (o + 1):D * (o + 2)]), bits(ref[o])), \
                f"T={T} order={o + 1} W={W}"
    b.close()


# ---- second set: R9, R10, R12, R13 as per-component operators (what the plugin's cEnergy, cMZcr, cAcf,
# cPitchACF, cDeltaRegression, cContourSmoother overrides call), against the oracle's functions
def _dev(capi, ctx, arr):
    import ctypes as C
    L = capi.load()
    p = C.c_void_p()
    assert L.smilehip_alloc(ctx._h, max(arr.nbytes, 8), C.byref(p)) == 0
    if arr.nbytes:
        assert L.smilehip_copy_to_device(ctx._h, p, arr.ctypes.data, arr.nbytes, None) == 0
    return p


def _host(capi, ctx, p, shape, dtype):
    L = capi.load()
    out = np.zeros(shape, dtype)
    assert L.smilehip_copy_to_host(ctx._h, out.ctypes.data, p, out.nbytes, None) == 0
    assert L.smilehip_stream_synchronize(ctx._h, None) == 0
    return out


def test_energy_zcr_acf_pitch_window_ops(oracle):
    import ctypes as C
    from opensmile_amd import capi
    ctx = capi.Context(0)
    L = capi.load()
    OL = oracle.lib()
    rng = np.random.default_rng(21)
    nF, N = 37, 400
    x = (rng.standard_normal((nF, N)) * 0.1).astype(np.float32)
    x[3] = 0.0
    x[4, ::2] = 0.0                                      # exact zeros exercise the (b == 0) branch of the ZCR test
    d_x = _dev(capi, ctx, x)
    # R12 sum of squares (double) and zero-crossing count
    d_d = _dev(capi, ctx, np.zeros(nF, np.float64))
    assert L.smilehip_sumsq_frames(ctx._h, d_x, N, N, nF, d_d, None) == 0
    d = _host(capi, ctx, d_d, nF, np.float64)
    OL.lldo_energy_rms.restype = C.c_float
    OL.lldo_energy_rms.argtypes = [C.c_void_p, C.c_long]
    OL.lldo_zcr.restype = C.c_float
    OL.lldo_zcr.argtypes = [C.c_void_p, C.c_long]
    for f in range(nF):
        rms = np.float32(np.sqrt(d[f] / np.float32(N)))
        assert abs(float(rms) - OL.lldo_energy_rms(x[f].ctypes.data, N)) <= 1e-7 * max(float(rms), 1e-30)
    d_c = _dev(capi, ctx, np.zeros(nF, np.int32))
    assert L.smilehip_zcr_count_frames(ctx._h, d_x, N, N, nF, d_c, None) == 0
    c = _host(capi, ctx, d_c, nF, np.int32)
    for f in range(nF):
        assert np.float32(c[f]) / np.float32(N) == np.float32(OL.lldo_zcr(x[f].ctypes.data, N))
    # R9 ACF and cepstrum of a magnitude spectrum (K = 257), then R10 on [acf | cepstrum]
    K = 257
    mag = np.abs(rng.standard_normal((nF, K))).astype(np.float32) + 0.01
    t = np.arange(K)
    mag += (2.0 + 2.0 * np.cos(2 * np.pi * t / 16.0)).astype(np.float32)    # harmonic structure -> a clear pitch peak
    cfg = capi.mfcc12_0_d_a_config()
    cfg.force_frame_size = 512
    cfg.stage_mask = 2                                   # SMILEHIP_STAGE_FFT
    cfg.n_delta = 0
    plan = capi.Plan(ctx, cfg)
    d_m = _dev(capi, ctx, mag)
    d_ac = _dev(capi, ctx, np.zeros((nF, 512), np.float32))
    assert L.smilehip_acf_frames(plan._h, d_m, K, d_ac, 512, 256, nF, 1, 0, 1, 0, None) == 0
    ac = _host(capi, ctx, d_ac, (nF, 512), np.float32)
    d_ce = C.c_void_p(d_ac.value + 256 * 4)
    assert L.smilehip_acf_frames(plan._h, d_m, K, d_ce, 512, 256, nF, 1, 1, 1, 0, None) == 0
    ac = _host(capi, ctx, d_ac, (nF, 512), np.float32)
    OL.lldo_acf.restype = None
    OL.lldo_acf.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_void_p]
    ref = np.zeros((nF, 512), np.float32)
    for f in range(nF):
        OL.lldo_acf(mag[f].ctypes.data, K, 0, ref[f].ctypes.data)
        OL.lldo_acf(mag[f].ctypes.data, K, 1, ref[f, 256:].ctypes.data)
    assert np.abs(ac - ref).max() <= 2e-6 * np.abs(ref).max()
    d_v = _dev(capi, ctx, np.zeros(nF, np.float64))
    d_i = _dev(capi, ctx, np.zeros(nF, np.int32))
    # fsSec is a float in cPitchACF (pitchACF.hpp): the caller hands over its double value
    assert L.smilehip_pitchacf_frames(ctx._h, d_ac, 512, 256, nF, float(np.float32(0.032)), 500.0, d_v, d_i, None) == 0
    v = _host(capi, ctx, d_v, nF, np.float64)
    idx = _host(capi, ctx, d_i, nF, np.int32)
    class St(C.Structure):
        _fields_ = [("lastPitch", C.c_float), ("lastlastPitch", C.c_float), ("glMeanPitch", C.c_float), ("onsFlag", C.c_int),
                    ("pitchEnv", C.c_float)]
    OL.lldo_pitch_acf.restype = None
    OL.lldo_pitch_acf.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_float, C.c_double, C.c_double, C.POINTER(St),
                                  C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    OL.lldo_pitch_state_init.argtypes = [C.POINTER(St)]
    Tsamp = float(np.float32(0.032)) / 512.0
    for f in range(nF):
        st = St()
        OL.lldo_pitch_state_init(C.byref(st))
        vp, f0, raw = C.c_float(), C.c_float(), C.c_float()
        OL.lldo_pitch_acf(ref[f].ctypes.data, ref[f, 256:].ctypes.data, 256, np.float32(0.032), 500.0, 0.0, C.byref(st),
                          C.byref(vp), C.byref(f0), C.byref(raw))
        # same inputs up to FFT round-off: the voicing probability agrees to 1e-5, the peak index exactly
        # (or, on a near-tie of two cepstral peaks, within the discontinuity the LLD tests bound)
        assert abs(np.float32(v[f]) - vp.value) <= 1e-5
        mine = np.float32(1.0) / (np.float32(idx[f]) * np.float32(Tsamp)) if idx[f] > 0 else np.float32(0.0)
        assert mine == np.float32(raw.value), (f, idx[f], mine, raw.value)
    # R13 one row through both window operators, valid on [-W, nT + W)
    nT, W = 1000, 2
    row = rng.standard_normal(nT + 2 * W).astype(np.float32)
    d_r = _dev(capi, ctx, row)
    d_y = _dev(capi, ctx, np.zeros(nT, np.float32))
    assert L.smilehip_window_op_row(ctx._h, C.c_void_p(d_r.value + 4 * W), d_y, nT, 0, W, None) == 0
    y = _host(capi, ctx, d_y, nT, np.float32)
    num = np.zeros(nT, np.float32)
    for i in (1, 2):
        num += np.float32(i) * (row[W + i:W + i + nT] - row[W - i:W - i + nT])
    assert np.array_equal(y, num / np.float32(10.0))
    assert L.smilehip_window_op_row(ctx._h, C.c_void_p(d_r.value + 4 * W), d_y, nT, 1, 1, None) == 0
    y = _host(capi, ctx, d_y, nT, np.float32)
    sma = row[W:W + nT].copy()
    sma += row[W - 1:W - 1 + nT]
    sma += row[W + 1:W + 1 + nT]
    assert np.array_equal(y, sma / np.float32(3.0))
    assert L.smilehip_window_op_row(ctx._h, d_r, d_y, nT, 2, 1, None) != 0       # unknown kind


def test_specscale_and_pitchshs_operators_bit_exact(oracle):
    """cSpecScale and cPitchShs as per-component operators on GIVEN inputs: everything after the FFT is ordered double /
    float arithmetic, so the device must reproduce the oracle bit for bit (rows: speech-like spectra, silence, a single
    peak, one with exactly one local maximum -- the reference's zero-initialised-list quirk)."""
    import ctypes as C
    from opensmile_amd import capi, synth
    ctx = capi.Context(0)
    L = capi.load()
    OL = oracle.lib()
    plan = capi.Plan(ctx, capi.compare16_f0_config())
    K = plan.geometry.n_bins
    assert K == 513
    # magnitude spectra of real frames (through the oracle's own front end) + special rows
    pcm = np.concatenate([synth.utterance(u, 8000) for u in (2, 3, 4, 10)])
    _, taps = oracle.compare_f0_chain(pcm, taps=True)
    rng = np.random.default_rng(5)
    mag = np.abs(rng.standard_normal((40, K))).astype(np.float32)
    t = np.arange(K)
    mag += (3.0 * (1 + np.cos(2 * np.pi * t / 13.0))).astype(np.float32)
    mag[0] = 0.0
    mag[1] = 0.0
    mag[1, 77] = 5.0
    mag[2] = np.linspace(1.0, 0.0, K, dtype=np.float32)          # exactly one local maximum (bin 0)
    mag[3] = np.float32(1.0)                                       # flat: no maximum at all

    class SS(C.Structure):
        _fields_ = [("K", C.c_long), ("ft", C.c_void_p), ("sigma", C.c_void_p), ("d1", C.c_void_p), ("d2", C.c_void_p),
                    ("k", C.c_void_p), ("co", C.c_void_p), ("audw", C.c_void_p), ("meta", C.c_float * 8)]

    class SH(C.Structure):
        _fields_ = [("N", C.c_long), ("n_octaves", C.c_float), ("points_per_octave", C.c_float), ("Fmint", C.c_float),
                    ("Fstept", C.c_float), ("base", C.c_double), ("n_harmonics", C.c_int), ("compression", C.c_float),
                    ("n_cand", C.c_int), ("min_pitch", C.c_double), ("max_pitch", C.c_double), ("voicing_cutoff", C.c_float)]
    ss, sh = SS(), SH()
    OL.lldo_specscale_init.restype = C.c_int
    OL.lldo_specscale_init.argtypes = [C.c_void_p, C.c_long, C.c_double]
    assert OL.lldo_specscale_init(C.byref(ss), K, plan.geometry.fft_frame_size_sec) == 1
    OL.lldo_shs_init.restype = None
    OL.lldo_shs_init.argtypes = [C.c_void_p, C.c_void_p]
    OL.lldo_shs_init(C.byref(sh), C.byref(ss))
    OL.lldo_specscale_frame.restype = None
    OL.lldo_specscale_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    OL.lldo_pitch_shs.restype = None
    OL.lldo_pitch_shs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    nF = mag.shape[0]
    ref_h = np.zeros((nF, K), np.float32)
    for f in range(nF):
        OL.lldo_specscale_frame(C.byref(ss), mag[f].ctypes.data, ref_h[f].ctypes.data)
    d_m = _dev(capi, ctx, mag)
    d_h = _dev(capi, ctx, np.zeros((nF, K), np.float32))
    assert L.smilehip_specscale_frames(plan._h, d_m, K, d_h, K, nF, None) == 0
    h = _host(capi, ctx, d_h, (nF, K), np.float32)
    assert np.array_equal(h.view(np.uint32), ref_h.view(np.uint32)), f"rows {sorted(set(np.argwhere(h != ref_h)[:, 0]))}"
    # cPitchShs on octave-scale spectra: the oracle's own level of real frames + the rows above
    hps = np.concatenate([taps["hps"], ref_h], axis=0)
    nH = hps.shape[0]
    ref_s = np.zeros((nH, 21), np.float32)
    for f in range(nH):
        OL.lldo_pitch_shs(C.byref(sh), hps[f].ctypes.data, ref_s[f].ctypes.data, None)
    d_hp = _dev(capi, ctx, hps)
    d_s = _dev(capi, ctx, np.zeros((nH, 21), np.float32))
    assert L.smilehip_pitchshs_frames(plan._h, d_hp, K, d_s, 21, nH, None) == 0
    s = _host(capi, ctx, d_s, (nH, 21), np.float32)
    same = (s.view(np.uint32) == ref_s.view(np.uint32)).all(axis=1)
    # exp() of the refined log2 frequency is the one libm call on the path (device vs glibc: <= 1 ulp in double,
    # visible after rounding to float on rare rows)
    assert same.mean() >= 0.99, f"{int((~same).sum())} of {nH} rows differ"
    assert np.abs(s - ref_s).max() <= 1e-6 * max(np.abs(ref_s).max(), 1.0)
    # error paths
    assert L.smilehip_specscale_frames(plan._h, d_m, K - 1, d_h, K, nF, None) != 0
    mf = capi.Plan(ctx, capi.mfcc12_0_d_a_config())
    assert L.smilehip_pitchshs_frames(mf._h, d_hp, K, d_s, 21, nH, None) != 0
