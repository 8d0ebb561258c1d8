"""The reference-order real FFT on the device (opensmile_amd/csrc/lld_ooura.hpp) against the oracle's rdft restatement
(oracle/lld_oracle_fft.c, itself pinned bit for bit against the REAL rdft in tests/test_ooura_fft.py): every word of the packed
spectrum must be identical -- zero signs included -- for every transform length the network is built for."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def own_rdft(x, isgn):
    from oracle import lldo
    L = lldo.lib()
    out = np.ascontiguousarray(x, dtype=np.float32).copy()
    fp = C.POINTER(C.c_float)
    for r in range(out.shape[0]):
        assert L.lldo_ooura_rdft(C.c_int(out.shape[1]), C.c_int(isgn), out[r].ctypes.data_as(fp)) == 0
    return out


def frames(n_frame, rows, seed):
    rng = np.random.default_rng(seed)
    x = (rng.integers(-32768, 32767, size=(rows, n_frame)).astype(np.float32) / np.float32(32767.0)).astype(np.float32)
    x[0] = 0.0
    x[1] = -0.0
    x[2] = 0.0; x[2, 1] = 1.0
    x[3] = np.where(np.arange(n_frame) % 160 < 80, 0.9, -0.9)
    x[4] = np.round(x[4] * 4) / 4
    x[5] = np.where(rng.random(n_frame) < 0.9, 0.0, x[5])
    x[6] = np.where(rng.random(n_frame) < 0.5, -0.0, 0.0)
    return x


@pytest.mark.parametrize("nfft,n_frame,sym", [(64, 64, 0), (128, 100, 1), (256, 200, 0), (512, 400, 0), (512, 320, 1),
                                              (1024, 960, 1), (1024, 1024, 0), (2048, 1103, 0), (4096, 2646, 1),
                                              (8192, 8192, 0)])
def test_rfft_stage_bits_equal_reference_order(nfft, n_frame, sym):
    import torch
    from opensmile_amd import capi
    ctx = capi.Context(0)
    cfg = capi.mfcc12_0_d_a_config()
    cfg.force_frame_size = n_frame
    cfg.zero_pad_symmetric = sym
    cfg.stage_mask = capi.STAGE_FFT
    plan = capi.Plan(ctx, cfg)
    assert plan.geometry.fft_size == nfft
    rows = 200 if nfft <= 1024 else 40
    x = frames(n_frame, rows, 31 + nfft + n_frame)
    d_x = torch.from_numpy(x).cuda()
    d_f = torch.empty((rows, nfft), dtype=torch.float32, device="cuda")
    capi.rfft_frames(plan, d_x.data_ptr(), n_frame, d_f.data_ptr(), nfft, rows)
    torch.cuda.synchronize()
    got = d_f.cpu().numpy()
    pad = (nfft - n_frame) // 2 if sym else 0
    padded = np.zeros((rows, nfft), dtype=np.float32)
    padded[:, pad:pad + n_frame] = x
    ref = own_rdft(padded, 1)
    diff = bits(got) != bits(ref)
    assert not diff.any(), f"Nfft={nfft}: {diff.sum()} of {diff.size} words differ, first at {np.argwhere(diff)[0]}"


@pytest.mark.parametrize("nfft", [128, 512, 1024, 2048])
def test_acf_stage_inverse_bits_equal_reference_order(nfft):
    """cAcf's inverse transform (rdft(N, -1) on the packed real spectrum, acf.cpp:308-343) through smilehip_acf_frames:
    |lag| values identical to the oracle's inverse network (pinned against the real rdft) -- FFT 512 / 1024 take the
    register form (one wave per frame), the other lengths the in-place LDS form."""
    import torch
    from opensmile_amd import capi
    ctx = capi.Context(0)
    cfg = capi.mfcc12_0_d_a_config()
    cfg.force_frame_size = nfft
    cfg.stage_mask = capi.STAGE_FFT
    plan = capi.Plan(ctx, cfg)
    K, M = nfft // 2 + 1, nfft // 2
    rows = 120
    rng = np.random.default_rng(5 + nfft)
    mag = np.abs(rng.standard_normal((rows, K))).astype(np.float32)
    mag[0] = 0.0
    mag[1] = 0.0; mag[1, 3] = 1.0
    mag[2] = np.round(mag[2] * 4) / 4
    d_m = torch.from_numpy(mag).cuda()
    d_a = torch.empty((rows, M), dtype=torch.float32, device="cuda")
    capi._check(capi.load().smilehip_acf_frames(plan._h, d_m.data_ptr(), K, d_a.data_ptr(), M, M, rows, 0, 0, 0, 0, None))
    torch.cuda.synchronize()
    got = d_a.cpu().numpy()
    packed = np.zeros((rows, nfft), dtype=np.float32)
    packed[:, 0] = mag[:, 0]
    packed[:, 1] = mag[:, K - 1]
    packed[:, 2::2] = mag[:, 1:K - 1]
    ref = np.abs(own_rdft(packed, -1)[:, :M])
    diff = bits(got) != bits(ref)
    assert not diff.any(), f"Nfft={nfft}: {diff.sum()} of {diff.size} words differ, first at {np.argwhere(diff)[0]}"
