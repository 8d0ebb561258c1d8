"""The C ABI's operators for the components' other option sets (smilehip_irfft_frames, smilehip_fftmagphase_frames,
smilehip_mzcr_frames, smilehip_delta_op_row) against numpy restatements of the reference lines they cite (transformFft.cpp:196-216 through the oracle's
rdft, fftmagphase.cpp:215-287, mzcr.cpp:108-150) on seeded rows, edge cases included. The plugin test
(tests/test_gpu_plugin.py::test_plugin_option_sets) holds the same operators against the real components."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def rows(n, rows_n, seed):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((rows_n, n)) * 0.3).astype(np.float32)
    x[0] = 0.0
    x[1] = -0.0
    x[2] = np.where(np.arange(n) % 7 < 3, 0.0, x[2])
    x[3] = np.abs(x[3])
    x[4] = -np.abs(x[4])
    return x


@pytest.mark.parametrize("nfft", [64, 512, 1024, 4096])
def test_irfft_frames_is_the_reference_inverse(nfft, oracle):
    import torch
    from opensmile_amd import capi
    ctx = capi.Context(0)
    cfg = capi.mfcc12_0_d_a_config()
    cfg.force_frame_size = nfft
    cfg.stage_mask = capi.STAGE_FFT
    plan = capi.Plan(ctx, cfg)
    x = rows(nfft, 40, 3 + nfft)
    d_x = torch.from_numpy(x).cuda()
    d_y = torch.empty_like(d_x)
    capi._check(capi.load().smilehip_irfft_frames(plan._h, d_x.data_ptr(), nfft, d_y.data_ptr(), nfft, len(x), None))
    torch.cuda.synchronize()
    L = oracle.lib()
    ref = x.copy()
    fp = C.POINTER(C.c_float)
    for r in range(len(ref)):
        assert L.lldo_ooura_rdft(C.c_int(nfft), C.c_int(-1), ref[r].ctypes.data_as(fp)) == 0
    ref = ref * (np.float32(2.0) / np.float32(nfft))
    assert np.array_equal(bits(d_y.cpu().numpy()), bits(ref))


def ref_magphase(a, flags, dbp_norm, min_dbp):
    f32 = np.float32
    n = a.shape[1]
    K = n // 2 + 1
    re = np.concatenate([a[:, :1], a[:, 2::2], a[:, 1:2]], axis=1).astype(f32)
    im = np.concatenate([np.zeros_like(a[:, :1]), a[:, 3::2], np.zeros_like(a[:, :1])], axis=1).astype(f32)
    edge = np.zeros(K, bool); edge[0] = edge[-1] = True
    fN = f32(n)
    out = []
    with np.errstate(divide="ignore", invalid="ignore"):
        if flags & 1:
            normalise, power, db = bool(flags & 4), bool(flags & 8), bool(flags & 16)
            sq = (re * re + im * im).astype(f32)
            mag = np.where(edge, np.abs(re), np.sqrt(sq)).astype(f32)
            if not db and not normalise and not power:
                m = mag
            elif not db and normalise and not power:
                m = (f32(1.0) / fN) * mag
            elif not db and normalise and power:
                e0 = (f32(1.0) / fN) * re
                e1 = (f32(1.0) / fN) * np.abs(re)
                m = (f32(1.0) / (fN * fN)) * sq
                m[:, 0] = (e0 * e0)[:, 0]
                m[:, -1] = (e1 * e1)[:, -1]
            elif not db and not normalise and power:
                m = np.where(edge, (np.abs(re) * np.abs(re)).astype(f32), sq)
            else:
                arg = np.where(edge, (f32(1.0) / fN) * np.abs(re), (f32(1.0) / (fN * fN)) * sq).astype(f32)
                lg = np.log10(arg.astype(np.float64)).astype(f32)            # (log10f: compared with a tolerance below)
                v = f32(dbp_norm) + np.where(edge, f32(20.0), f32(10.0)) * lg
                m = np.maximum(f32(min_dbp), v)
            out.append(m.astype(f32))
        if flags & 2:
            ph = np.arctan2(im.astype(np.float64), re.astype(np.float64)).astype(f32)
            ph = np.where(edge, np.where(re >= 0, f32(0), f32(np.pi)), ph)
            out.append(ph.astype(f32))
    return np.concatenate(out, axis=1)


@pytest.mark.parametrize("flags", [1, 1 | 4, 1 | 8, 1 | 4 | 8, 1 | 16, 2, 1 | 2, 1 | 2 | 16])
def test_fftmagphase_every_mode(flags):
    import torch
    from opensmile_amd import capi
    ctx = capi.Context(0)
    nfft = 512
    a = rows(nfft, 64, 77 + flags)
    K = nfft // 2 + 1
    n_out = (K if flags & 1 else 0) + (K if flags & 2 else 0)
    d_a = torch.from_numpy(a).cuda()
    d_o = torch.full((len(a), n_out), float("nan"), dtype=torch.float32, device="cuda")
    capi._check(capi.load().smilehip_fftmagphase_frames(ctx._h, d_a.data_ptr(), nfft, nfft, flags, 90.302, -29.698, d_o.data_ptr(), n_out,
                                                        len(a), None))
    torch.cuda.synchronize()
    got = d_o.cpu().numpy()
    ref = ref_magphase(a, flags, 90.302, -29.698)
    if flags & 16 or flags & 2:          # log10f / atan2f: the float functions against float64-then-rounded values
        both = np.isfinite(ref) & np.isfinite(got)
        assert (np.isfinite(ref) == np.isfinite(got)).all()
        assert np.abs(got[both] - ref[both]).max() <= 2e-5
        if not flags & 2 or flags & 1:
            exact = slice(0, K) if not flags & 16 else slice(0, 0)
            assert np.array_equal(bits(got[:, exact]), bits(ref[:, exact]))
    else:
        assert np.array_equal(bits(got), bits(ref))


def ref_mzcr(x, flags):
    f32 = np.float32
    out = []
    for r in x:
        N = len(r)
        mean = r[0]
        for i in range(1, N - 1):
            mean = f32(mean + r[i])
        mean = f32(mean / f32(N))
        a, b, c = r[:-2], r[1:-1], r[2:]
        nz = int((((a * c <= 0) & (b == 0)) | (a * b < 0)).sum())
        am, bm, cm = (a - mean).astype(f32), (b - mean).astype(f32), (c - mean).astype(f32)
        nm = int((((am * cm <= 0) & (bm == 0)) | (am * bm < 0)).sum())
        row = []
        if flags & 1: row.append(f32(nz) / f32(N))
        if flags & 2: row.append((f32(4.0) + f32(nm)) / f32(N))
        mx, mn = r.max(), r.min()
        if flags & 4: row.append(max(abs(mn), abs(mx)) if abs(mn) > abs(mx) else abs(mx))
        if flags & 8: row += [mx, mn]
        if flags & 16: row.append(mean)
        out.append(row)
    return np.array(out, f32)


@pytest.mark.parametrize("N,flags", [(400, 31), (160, 2), (1103, 1 | 4 | 16), (3, 31), (2, 8 | 16), (1, 4)])
def test_mzcr_every_output(N, flags):
    import torch
    from opensmile_amd import capi
    ctx = capi.Context(0)
    x = rows(max(N, 8), 48, 5 + N)[:, :N].copy()
    n_out = sum(((flags >> b) & 1) * w for b, w in ((0, 1), (1, 1), (2, 1), (3, 2), (4, 1)))
    d_x = torch.from_numpy(x).cuda()
    d_o = torch.full((len(x), n_out), float("nan"), dtype=torch.float32, device="cuda")
    capi._check(capi.load().smilehip_mzcr_frames(ctx._h, d_x.data_ptr(), N, N, len(x), flags, d_o.data_ptr(), n_out, None))
    torch.cuda.synchronize()
    got, ref = d_o.cpu().numpy(), ref_mzcr(x, flags)
    same = (bits(got) == bits(ref)) | ((got == 0) & (ref == 0))
    assert same.all(), (np.argwhere(~same)[:5], got[~same][:5], ref[~same][:5])


def _delta_ref(x, pre, nT, W, flags, norm0):
    """cDeltaRegression::processBuffer (deltaRegression.cpp:104-170) in float32, the reference's statement order; x[pre + n] = sample n"""
    f = np.float32
    rel, half, ab, seg = flags & 1, flags & 2, flags & 4, flags & 8
    no = lambda v: v == 0.0 or v != v

    def delta(prior, later):
        d = f(later - prior)
        if rel:
            d = f(d / f(abs(prior))) if prior != 0.0 else f(0.0)
        return d
    y = np.zeros(nT, np.float32)
    norm = f(norm0)
    for n in range(nT):
        c = pre + n
        if W > 0:
            num = f(0.0)
            for i in range(1, W + 1):
                if seg and (no(x[c + i]) or no(x[c - i])):
                    continue
                num = f(num + f(f(i) * delta(x[c - i], x[c + i])))
                if seg:
                    norm = f(norm + f(f(i) * f(i)))
            y[n] = f(num / norm) if (not seg or norm != 0.0) else f(0.0)
        else:
            y[n] = f(0.0) if (seg and (no(x[c]) or no(x[c - 1]))) else delta(x[c - 1], x[c])
    if half:
        y[y < 0.0] = 0.0
    elif ab:
        y = np.where(y < 0.0, -y, y).astype(np.float32)
    return y, norm


@pytest.mark.parametrize("W", [0, 1, 2, 3])
@pytest.mark.parametrize("flags", [0, 1, 2, 4, 6, 5, 8, 9, 13, 11])
def test_delta_op_row_every_option(W, flags):
    import torch
    from opensmile_amd import capi
    ctx = capi.Context(0)
    L = capi.load()
    rng = np.random.default_rng(100 * W + flags)
    nT, pre = 97, max(W, 1)
    x = (rng.standard_normal(nT + pre + W) * 0.5).astype(np.float32)
    x[rng.random(len(x)) < 0.25] = 0.0                      # "no value" stretches (unvoiced frames of an F0 contour)
    x[10] = np.nan
    norm0 = np.float32(2.0) * np.float32(sum(i * i for i in range(1, W + 1)))
    d_x = torch.from_numpy(x).cuda()
    d_y = torch.zeros(nT, dtype=torch.float32, device="cuda")
    d_norm = torch.tensor([norm0], dtype=torch.float32, device="cuda")
    ref_all, got_all = [], []
    norm = norm0
    for rep in range(2):                                    # two rows in the instance's order: onlyInSegments carries its divisor
        ref, norm = _delta_ref(x, pre, nT, W, flags, norm)
        capi._check(L.smilehip_delta_op_row(ctx._h, d_x.data_ptr() + 4 * pre, d_y.data_ptr(), nT, W, flags, d_norm.data_ptr(), None))
        torch.cuda.synchronize()
        ref_all.append(ref)
        got_all.append(d_y.cpu().numpy().copy())
    for ref, got in zip(ref_all, got_all):
        same = (bits(ref) == bits(got)) | (np.isnan(ref) & np.isnan(got))
        assert same.all(), f"W={W} flags={flags}: {np.argwhere(~same)[:5].ravel()} {got[~same][:3]} vs {ref[~same][:3]}"
    if (flags & 8) and W > 0:
        assert bits(d_norm.cpu().numpy())[0] == bits(np.array([norm]))[0]


@pytest.mark.parametrize("K,min_f", [(513, 25.0), (257, 20.0), (2049, 25.0)])
@pytest.mark.parametrize("off", [0, 1, 2, 4, 7, 5])
def test_specscale_switches(K, min_f, off, oracle):
    """cSpecScale with specEnhance / specSmooth / auditoryWeighting switched off one by one and together (emobase2010:
    all three off; IS10_paraling_compat likewise): smilehip_specscale_frames on a plan with specscale_off against the
    oracle's lldo_specscale_frame_ex, which tests/test_oracle_pin_f0_variants.py pins on those files' own levels. Without the
    weighting the spline's negative values pass."""
    import torch
    from opensmile_amd import capi
    ctx = capi.Context(0)
    cfg = capi.compare16_f0_config()
    fs = (K - 1) * 2 / 16000.0
    cfg.force_fft_frame_size_sec = fs
    cfg.force_frame_size = 2 * (K - 1)
    cfg.specscale_min_f = min_f
    cfg.specscale_off = off
    plan = capi.Plan(ctx, cfg)
    assert plan.geometry.n_bins == K
    rng = np.random.default_rng(K + off)
    mag = np.abs(rng.standard_normal((12, K))).astype(np.float32)
    mag[0] = 0.0
    mag[1] = np.linspace(1.0, 0.0, K, dtype=np.float32)
    mag[2] = 1.0
    mag[3, ::7] *= 40.0                                      # sharp peaks: the spline overshoots below zero between them
    ref, _ = oracle.specscale_shs_rows(mag, fs, min_f=min_f, flags=7 & ~off)
    d_m = torch.from_numpy(mag).cuda()
    d_h = torch.zeros((12, K), dtype=torch.float32, device="cuda")
    capi._check(capi.load().smilehip_specscale_frames(plan._h, d_m.data_ptr(), K, d_h.data_ptr(), K, 12, None))
    torch.cuda.synchronize()
    got = d_h.cpu().numpy()
    d = bits(got) != bits(ref)
    assert not d.any(), f"K={K} off={off}: {d.sum()} cells differ, rows {sorted(set(np.argwhere(d)[:, 0]))}"
    if off & 4:
        assert (ref[3] < 0).any()
    # the fused chain refuses a plan with switches off
    if off:
        b = capi.Batch(plan, np.array([0, 16000], np.int64))
        with pytest.raises(Exception):
            b.run_host(np.zeros(16000, np.int16))
