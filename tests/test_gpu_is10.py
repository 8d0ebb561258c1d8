"""The operators for the components the other INTERSPEECH sets add (cIntensity, cLsp, cPitchSmoother, cVectorOperation, cSpecResample
and cLpc for any geometry; lld_stage4_kernels.hip) and the Onset functional family, through the C ABI against the oracle on seeded
rows -- the oracle itself is pinned on the real binary's levels (tests/test_oracle_pin_is10.py), and the kernels' per-frame bodies
are the host-checked lld_is10_ops.hpp. The plugin test (tests/test_gpu_plugin.py::test_plugin_is10_paraling) holds the same operators
against the real components inside the reference binary."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits_equal(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))))


@pytest.fixture(scope="module")
def env():
    import torch
    from opensmile_amd import capi
    return torch, capi, capi.Context(0)


def test_intensity_lsp_vecop(env, oracle):
    torch, capi, ctx = env
    L = capi.load()
    rng = np.random.default_rng(21)
    fr = (rng.standard_normal((300, 400)) * 0.2).astype(np.float32)
    fr[0] = 0.0
    d = torch.from_numpy(fr).cuda()
    for flags, (i, l) in ((1, (1, 0)), (2, (0, 1)), (3, (1, 1))):
        w = bin(flags).count("1")
        out = torch.full((len(fr), w), 9.0, device="cuda")
        capi._check(L.smilehip_intensity_frames(ctx._h, d.data_ptr(), 400, 400, flags, out.data_ptr(), w, len(fr), None))
        torch.cuda.synchronize()
        assert bits_equal(out.cpu().numpy(), oracle.intensity_rows(fr, i, l)), flags
    for p in (8, 10, 16, 32):
        lpc = []
        for _ in range(500):
            k = rng.uniform(-0.95, 0.95, p)
            a = np.zeros(0)
            for m in range(p):
                a = np.concatenate([a + k[m] * a[::-1], [k[m]]])
            lpc.append(a)
        lpc = np.concatenate([np.array(lpc, np.float32), (rng.standard_normal((50, p)) * 2).astype(np.float32), np.zeros((2, p), np.float32)])
        dl = torch.from_numpy(lpc).cuda()
        out = torch.full_like(dl, 9.0)
        capi._check(L.smilehip_lsp_frames(ctx._h, dl.data_ptr(), p, p, out.data_ptr(), p, len(lpc), None))
        torch.cuda.synchronize()
        assert bits_equal(out.cpu().numpy(), oracle.lsp_rows(lpc)), p
    v = np.concatenate([rng.standard_normal(4089) * 3, [0.0, -0.0, 1e-13, 1e-12, 1e-30, 88.0, -104.0]]).astype(np.float32).reshape(-1, 8)
    dv = torch.from_numpy(v).cuda()
    for op, p1 in (("add", 0.37), ("mul", -2.5), ("log", 1.0), ("lgA", 10.0), ("sqr", 1.0), ("ee", 1.0), ("abs", 1.0), ("dBp", 1.0), ("dBv", 1.0)):
        out = torch.empty_like(dv)
        capi._check(L.smilehip_vecop_frames(ctx._h, oracle.VOP[op], p1, 0.0, dv.data_ptr(), 8, 8, out.data_ptr(), 8, v.shape[0], None))
        torch.cuda.synchronize()
        assert bits_equal(out.cpu().numpy(), oracle.vecop_rows(v, op, p1)), op
    m = (rng.standard_normal((200, 26)) * rng.uniform(0.01, 30, (200, 1))).astype(np.float32)
    m[0] = 0.0
    dm = torch.from_numpy(m).cuda()
    for op in ("sum", "ssm", "ll1", "ll2"):                 # the vector-to-scalar operations (ComParE_2016's audspec sums are ll1)
        out = torch.full((len(m), 1), 9.0, device="cuda")
        capi._check(L.smilehip_vecop_frames(ctx._h, oracle.VOP[op], 1.0, 0.0, dm.data_ptr(), 26, 26, out.data_ptr(), 1, len(m), None))
        torch.cuda.synchronize()
        assert bits_equal(out.cpu().numpy()[:, 0], oracle.vecop_reduce_rows(m, op)), op


def test_pitch_smoother_streams_and_resume(env, oracle):
    from test_is10_ops_host import random_cands
    torch, capi, ctx = env
    L = capi.load()
    rng = np.random.default_rng(23)
    lens = [0, 1, 2, 57, 400, 1, 1203]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    for c in (6, 3):
        x = random_cands(rng, int(off[-1]), c)
        dx = torch.from_numpy(x).cuda()
        doff = torch.from_numpy(off).cuda()
        for octc, simple, flags in ((0, 1, 1), (0, 1, 2 | 8), (1, 1, 15), (1, 0, 3), (0, 0, 4 | 8)):
            w = bin(flags).count("1")
            out = torch.full((len(x), w), 9.0, device="cuda")
            wr = torch.zeros(len(lens), dtype=torch.int64, device="cuda")
            capi._check(L.smilehip_pitch_smoother_rows(ctx._h, c, 0.7, octc, simple, flags, dx.data_ptr(), 3 * c, doff.data_ptr(), len(lens), 0,
                                                       None, 0, out.data_ptr(), w, wr.data_ptr(), None))
            torch.cuda.synchronize()
            o, wrote = out.cpu().numpy(), wr.cpu().numpy()
            for u, n in enumerate(lens):
                ref = oracle.pitch_smoother_rows(x[off[u]:off[u + 1]], c, 0.7, octc, simple, flags)
                assert wrote[u] == len(ref) == (max(n - 1, 0) if (simple and flags & 3) else n), (u, n, wrote[u])
                assert bits_equal(o[off[u]:off[u] + len(ref)], ref), (c, octc, simple, flags, u)
        # one stream pushed frame by frame with the carried state (what the plugin does)
        x1 = x[:300]
        ref = oracle.pitch_smoother_rows(x1, c, 0.7, 1, 1, 15)
        state = torch.zeros(8, dtype=torch.int32, device="cuda")
        got = []
        row = torch.empty((1, 3 * c), device="cuda")
        o4 = torch.empty((1, 4), device="cuda")
        wr = torch.zeros(1, dtype=torch.int64, device="cuda")
        for t in range(len(x1)):
            row.copy_(torch.from_numpy(x1[t:t + 1]))
            capi._check(L.smilehip_pitch_smoother_rows(ctx._h, c, 0.7, 1, 1, 15, row.data_ptr(), 3 * c, None, 1, 1, state.data_ptr(),
                                                       1 if t else 0, o4.data_ptr(), 4, wr.data_ptr(), None))
            torch.cuda.synchronize()
            if int(wr.cpu()[0]) > 0:
                got.append(o4.cpu().numpy()[0].copy())
        assert bits_equal(np.array(got, np.float32), ref)


@pytest.mark.parametrize("n_in,n_frame,rate,target", [(512, 400, 16000, 11000), (512, 320, 16000, 11000), (1024, 1024, 16000, 8000),
                                                      (256, 200, 8000, 11000), (2048, 1102, 44100, 11000)])
def test_specresample_and_lpc_any_geometry(env, oracle, n_in, n_frame, rate, target):
    torch, capi, ctx = env
    L = capi.load()
    rng = np.random.default_rng(n_in + n_frame)
    spec = (rng.standard_normal((61, n_in)) * 0.5).astype(np.float32)       # 15 blocks of four frames + one
    spec[0] = 0.0
    fs_sec, last, bp = n_in / rate, n_frame / rate, 1.0 / rate
    n_out, k_max, nd = C.c_int64(0), C.c_int64(0), C.c_double(0.0)
    capi._check(L.smilehip_specresample_geometry(n_in, fs_sec, last, bp, float(target), C.byref(n_out), C.byref(k_max), C.byref(nd)))
    ref = oracle.specresample_rows(spec, fs_sec, last, bp, float(target))
    assert ref.shape[1] == n_out.value
    h = k_max.value // 2
    ct, st = np.zeros(h * n_out.value, np.float32), np.zeros(h * n_out.value, np.float32)
    capi._check(L.smilehip_specresample_tables(n_in, n_out.value, k_max.value, nd.value, ct.ctypes.data, st.ctypes.data))
    d_ct, d_st, d_in = torch.from_numpy(ct).cuda(), torch.from_numpy(st).cuda(), torch.from_numpy(spec).cuda()
    out = torch.full((len(spec), n_out.value), 9.0, device="cuda")
    capi._check(L.smilehip_specresample_table_frames(ctx._h, d_in.data_ptr(), n_in, n_in, n_out.value, k_max.value, d_ct.data_ptr(),
                                                     d_st.data_ptr(), out.data_ptr(), n_out.value, len(spec), None))
    torch.cuda.synchronize()
    y = out.cpu().numpy()
    assert bits_equal(y, ref)
    for p in (8, 11, 16, 32):
        lp = torch.full((len(spec), p), 9.0, device="cuda")
        capi._check(L.smilehip_lpc_acf_frames(ctx._h, out.data_ptr(), n_out.value, n_out.value, p, lp.data_ptr(), p, len(spec), None))
        torch.cuda.synchronize()
        assert bits_equal(lp.cpu().numpy(), oracle.egemaps_lpc_rows(y, p)), p


def test_onset_family(env, oracle):
    torch, capi, ctx = env
    L = capi.load()
    rng = np.random.default_rng(5)
    for rows in (1, 2, 9, 300, 1500):
        x = (rng.standard_normal((rows, 7)) * (rng.random((rows, 7)) > 0.4)).astype(np.float32)
        dx = torch.from_numpy(x).cuda()
        for norm in ("segment", "second", "frame"):
            for use_abs, th_on, th_off in ((0, 0.0, 0.0), (1, 0.3, 0.3), (0, 0.5, -0.2), (1, 0.1, 0.6)):
                so = oracle.FuncSpec()
                oracle._spec_common(so, ["Onset", "Times"])
                so.ons_mask, so.ons_norm, so.ons_use_abs = 0x1f, oracle.NORM[norm], use_abs
                so.ons_thr_on, so.ons_thr_off = th_on, th_off
                so.times_mask, so.times_norm = 1 << 12, oracle.NORM["second"]
                s = capi.FuncSpec()
                C.memmove(C.byref(s), C.byref(so), C.sizeof(s))
                per = capi.funcspec_count(s)
                assert per == 6
                out = torch.full((7 * per,), 9.0, device="cuda")
                capi._check(L.smilehip_funcspec_matrix(ctx._h, C.byref(s), dx.data_ptr(), 7, rows, 7, out.data_ptr(), None))
                torch.cuda.synchronize()
                ref = oracle.funcspec(x, so)
                got = out.cpu().numpy().reshape(7, per)
                assert bits_equal(got, ref), (rows, norm, use_abs, th_on, th_off)


def test_peaks_family(env, oracle):
    """The older peak picker (functionalPeaks.cpp, IS11_speaker_state's "Peaks") on the GPU = the oracle (pinned on the binary)."""
    torch, capi, ctx = env
    L = capi.load()
    rng = np.random.default_rng(9)
    for rows in (2, 3, 10, 298, 2000):
        x = np.cumsum(rng.standard_normal((rows, 9)), axis=0).astype(np.float32)
        x[:, 1] = np.round(x[:, 1])                         # plateaus and ties
        x[:, 2] = 0.0
        x[:, 3] = np.sin(np.arange(rows) * 0.3).astype(np.float32) * (1 + 0.3 * rng.standard_normal(rows).astype(np.float32))
        dx = torch.from_numpy(x).cuda()
        for norm in ("segment", "second", "frame"):
            so = oracle.FuncSpec()
            oracle._spec_common(so, ["Peaks", "Extremes"])
            so.pko_mask, so.pko_norm = 0x1f, oracle.NORM[norm]
            so.ext_mask, so.ext_norm = 0x7, oracle.NORM["frame"]
            s = capi.FuncSpec()
            C.memmove(C.byref(s), C.byref(so), C.sizeof(s))
            per = capi.funcspec_count(s)
            assert per == 8
            out = torch.full((9 * per,), 9.0, device="cuda")
            capi._check(L.smilehip_funcspec_matrix(ctx._h, C.byref(s), dx.data_ptr(), 9, rows, 9, out.data_ptr(), None))
            torch.cuda.synchronize()
            assert bits_equal(out.cpu().numpy().reshape(9, per), oracle.funcspec(x, so)), (rows, norm)


def test_crossings_dct_samples_families(env, oracle):
    """The families Crossings, DCT (its cosines formed on the device) and Samples on the GPU = the oracle (pinned on the binary)."""
    torch, capi, ctx = env
    L = capi.load()
    rng = np.random.default_rng(13)
    for rows in (1, 2, 3, 57, 298, 1000):
        x = (rng.standard_normal((rows, 6)) * 2).astype(np.float32)
        x[:, 1] = np.round(x[:, 1])
        x[:, 2] = 0.0
        x[:, 3] += 5.0
        dx = torch.from_numpy(x).cuda()
        so = oracle.FuncSpec()
        oracle._spec_common(so, ["Crossings", "DCT", "Samples"])
        so.crs_mask, so.dct_first, so.dct_last, so.n_samples = 7, 0, 6, 5
        for i, p in enumerate([0.0, 0.33, 0.5, 0.999, 1.0]):
            so.sample_pos[i] = p
        s = capi.FuncSpec()
        C.memmove(C.byref(s), C.byref(so), C.sizeof(s))
        per = capi.funcspec_count(s)
        assert per == 3 + 7 + 5
        out = torch.full((6 * per,), 9.0, device="cuda")
        capi._check(L.smilehip_funcspec_matrix(ctx._h, C.byref(s), dx.data_ptr(), 6, rows, 6, out.data_ptr(), None))
        torch.cuda.synchronize()
        got, ref = out.cpu().numpy().reshape(6, per), oracle.funcspec(x, so)
        assert bits_equal(got[:, :3], ref[:, :3]) and bits_equal(got[:, 10:], ref[:, 10:]), rows
        # DCT: the device's double cosine and libm's agree after the rounding to float but for ~1e-9 of the table entries
        d = got[:, 3:10].view(np.uint32) != ref[:, 3:10].view(np.uint32)
        assert d.sum() <= 1 and np.allclose(got[:, 3:10], ref[:, 3:10], rtol=1e-5, atol=1e-6), (rows, d.sum())
