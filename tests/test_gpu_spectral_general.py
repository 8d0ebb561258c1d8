"""R11, general option set on the device (smilehip_spectral_op_*, csrc/lld_spectral_general.hip) against the oracle's
lldo_spectral_general, which tests/test_oracle_pin_spectral_sets.py pins bit for bit on the real binary's own cSpectral levels of
avec2011, emo_large and the MediaEval files: the same three option sets, a few more (every flag alone, no band, sixteen bands),
spectra of speech-like frames plus an all-zero and a constant one, the frames of a stream in one launch and frame by frame
(d_state carrying the flux's previous frame) -- the same bits either way."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SETS = {
    "avec2011": ([(250, 650), (1000, 4000)], dict(flux=1, entropy=1, variance=1, skewness=1, kurtosis=1, sharpness=1, harmonicity=1)),
    "emo_large": ([(0, 250), (0, 650), (250, 650), (1000, 4000)], dict(flux=1, centroid=1, max_pos=1, min_pos=1)),
    "mediaeval": ([(40, 150), (250, 650), (1000, 4000), (5000, 15000)],
                  dict(flux=1, centroid=1, entropy=1, variance=1, skewness=1, kurtosis=1, slope=1, harmonicity=1, sharpness=1)),
    "all": ([(0, 100), (100, 8000), (7999, 8000), (300, 301)],
            dict(flux=1, centroid=1, max_pos=1, min_pos=1, entropy=1, variance=1, skewness=1, kurtosis=1, slope=1, sharpness=1, harmonicity=1)),
    "avec2013": ([(250, 650), (1000, 4000)], dict(flux=1, entropy=1, variance=1, skewness=1, kurtosis=1, sharpness=1, harmonicity=1, flatness=1)),
    "log_flatness": ([(250, 650)], dict(flatness=1, log_flatness=1)),
    "no_band_slope": ([], dict(slope=1)),
    "sixteen": ([(100 * i, 100 * i + 450) for i in range(16)], dict(kurtosis=1)),
    "flux_only": ([], dict(flux=1)),
    # round 6: the options no shipped file uses (pinned on the real binary by tests/test_oracle_pin_spectral_sets.py::test_spectral_round6_options_bit_exact)
    "r6_all": ([(250, 650), (1000, 4000)], dict(spec_diff=1, spec_pos_diff=1, flux=1, flux_centroid=1, flux_at_flux_centroid=1, centroid=1,
                                               standard_deviation=1, variance=1, skewness=1)),
    "r6_noflux": ([(0, 250)], dict(spec_diff=1, flux_centroid=1, standard_deviation=1)),
    "r6_posdiff": ([], dict(spec_pos_diff=1, flux_at_flux_centroid=1, kurtosis=1, entropy=1)),
    "r6_slopes": ([], dict(slope=1)),
}
SLOPES = {"r6_all": [(0, 500), (500, 1500), (1500, 8000)], "r6_noflux": [(300, 3400)], "r6_slopes": [(0, 8000), (100, 101), (7000, 7999)]}


def _spectra(K, n, seed):
    from opensmile_amd import synth
    N = (K - 1) * 2
    x = synth.utterance(seed, N + 160 * (n - 1)).astype(np.float32) / np.float32(32767.0)
    w = np.hamming(N).astype(np.float32)
    rows = [np.abs(np.fft.rfft((x[160 * t:160 * t + N] * w).astype(np.float64))).astype(np.float32) for t in range(n)]
    mag = np.stack(rows)
    mag[3] = 0.0
    mag[4] = 1.0
    return np.ascontiguousarray(mag)


@pytest.mark.parametrize("name", sorted(SETS))
@pytest.mark.parametrize("K", [257, 513])
def test_spectral_general_equals_oracle(name, K, oracle):
    import torch
    from opensmile_amd import capi
    bands, flags = SETS[name]
    slopes = SLOPES.get(name, [])
    rolloff = (0.25, 0.5, 0.75, 0.9) if name != "flux_only" else ()
    ctx = capi.Context(0)
    L = capi.load()
    o = capi.spectral_opts(bands, rolloff, slopes, **flags)
    fs = (K - 1) * 2 / 16000.0
    op = C.c_void_p()
    capi._check(L.smilehip_spectral_op_create(ctx._h, C.byref(o), K, fs, C.byref(op)))
    n_out = L.smilehip_spectral_op_n_out(op)
    assert n_out == L.smilehip_spectral_opts_count(C.byref(o)) == len(bands) + len(slopes) + len(rolloff) + sum(v for k, v in flags.items() if k != "log_flatness")
    mag = _spectra(K, 24, 11 if K == 257 else 12)
    ref = oracle.spectral_general_rows(mag, fs, bands, rolloff, slopes=slopes, **flags)
    assert ref.shape == (24, n_out)
    d_mag = torch.from_numpy(mag).cuda()
    d_state = torch.zeros(K, dtype=torch.float32, device="cuda")
    d_out = torch.full((24, n_out + 3), float("nan"), dtype=torch.float32, device="cuda")
    # the whole stream in one launch
    capi._check(L.smilehip_spectral_op_frames(op, d_mag.data_ptr(), K, d_state.data_ptr(), 1, d_out.data_ptr(), n_out + 3, 24, None))
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    assert np.isnan(got[:, n_out:]).all()
    d = got[:, :n_out].view(np.uint32) != ref.view(np.uint32)
    assert not d.any(), f"{name} K={K}: {d.sum()} cells differ, columns {sorted(set(np.argwhere(d)[:, 1]))}, rows {sorted(set(np.argwhere(d)[:, 0]))[:6]}"
    # frame by frame and in uneven pieces: the state carries the previous frame
    d_state.zero_()
    d_out2 = torch.zeros((24, n_out), dtype=torch.float32, device="cuda")
    t = 0
    for piece in (1, 1, 5, 2, 15):
        capi._check(L.smilehip_spectral_op_frames(op, d_mag[t:].data_ptr(), K, d_state.data_ptr(), 1 if t == 0 else 0,
                                                  d_out2[t:].data_ptr(), n_out, piece, None))
        t += piece
    torch.cuda.synchronize()
    assert np.array_equal(d_out2.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    capi._check(L.smilehip_spectral_op_destroy(op))


def test_spectral_general_argument_checks():
    from opensmile_amd import capi
    ctx = capi.Context(0)
    L = capi.load()
    op = C.c_void_p()
    bad = capi.spectral_opts([(650, 250)], (), flux=1)
    assert L.smilehip_spectral_op_create(ctx._h, C.byref(bad), 257, 0.032, C.byref(op)) != 0
    none = capi.spectral_opts([], ())
    assert L.smilehip_spectral_opts_count(C.byref(none)) == 0 and L.smilehip_spectral_op_create(ctx._h, C.byref(none), 257, 0.032, C.byref(op)) != 0
    ok = capi.spectral_opts([(250, 650)], (0.5,), flux=1)
    capi._check(L.smilehip_spectral_op_create(ctx._h, C.byref(ok), 257, 0.032, C.byref(op)))
    assert L.smilehip_spectral_op_frames(op, None, 257, None, 1, None, 3, 0, None) != 0        # flux without a state buffer
    capi._check(L.smilehip_spectral_op_destroy(op))
