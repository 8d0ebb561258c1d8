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
