"""smilehip_melspec_inverse_table_frames (cMelspec with inverse = 1, melspec.cpp:466-516) against the oracle, bit for bit, through the C ABI:
seeded mel-band rows (positive, zero and negative bands: the square root's branch), the oracle's own tables uploaded as the plugin uploads
the component's, rows with a leading dimension wider than the row, an empty call, and the reference's golden chain (the bands the real
binary wrote for a seeded utterance: tests/test_oracle_pin_melspec_inverse.py pins the oracle on its spectrum)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = {"power_htk_257": (26, 257, 1, 1, 0.0, 8000.0), "mag_257": (26, 257, 0, 1, 0.0, 8000.0), "power_nohtk_129": (26, 129, 1, 0, 0.0, 8000.0),
         "power_htk_band": (26, 257, 1, 1, 300.0, 6000.0), "forty_bands_513": (40, 513, 1, 1, 20.0, 7600.0)}


@pytest.mark.parametrize("case", sorted(CASES))
def test_melspec_inverse_equals_oracle(case, oracle):
    import torch
    from opensmile_amd import capi
    n_src, K, power, htk, lo, hi = CASES[case]
    fss = (K - 1) * 2 / 16000.0
    rng = np.random.default_rng(7 + K + n_src)
    mel = (rng.standard_normal((37, n_src)) * (3.0e8 if (htk and power) else 4.0e3)).astype(np.float32)
    mel[:30] = np.abs(mel[:30])                          # the usual case: band energies; the last rows keep negative bands
    mel[5] = 0.0
    ref, (n_lo, n_hi, coef, chan) = oracle.melspec_inverse_rows(mel, K, fss, lo, hi, power, htk, tables=True)
    assert np.abs(ref).max() > 0 and (ref[30:] == 0).any()
    ctx = capi.Context(0)
    L = capi.load()
    d_coef, d_chan = torch.from_numpy(coef).cuda(), torch.from_numpy(chan).cuda()
    ld_src, ld_dst = n_src + 2, K + 5
    src = np.full((37, ld_src), np.nan, np.float32)
    src[:, :n_src] = mel
    d_src = torch.from_numpy(src).cuda()
    d_dst = torch.full((37, ld_dst), float("nan"), dtype=torch.float32, device="cuda")
    div = (32767.0 * 32767.0 if power else 32767.0) if htk else 1.0
    capi._check(L.smilehip_melspec_inverse_table_frames(ctx._h, d_src.data_ptr(), ld_src, n_src, K, d_coef.data_ptr(), d_chan.data_ptr(), n_lo, n_hi,
                                                        power, np.float32(div).item(), d_dst.data_ptr(), ld_dst, 37, None))
    torch.cuda.synchronize()
    got = d_dst.cpu().numpy()
    assert np.isnan(got[:, K:]).all()
    d = got[:, :K].view(np.uint32) != ref.view(np.uint32)
    assert not d.any(), f"{case}: {d.sum()} cells differ, first {np.argwhere(d)[:4].tolist()}"
    # no frames: nothing is touched; bad arguments are refused
    capi._check(L.smilehip_melspec_inverse_table_frames(ctx._h, None, ld_src, n_src, K, d_coef.data_ptr(), d_chan.data_ptr(), n_lo, n_hi, power,
                                                        np.float32(div).item(), None, ld_dst, 0, None))
    assert L.smilehip_melspec_inverse_table_frames(ctx._h, d_src.data_ptr(), n_src - 1, n_src, K, d_coef.data_ptr(), d_chan.data_ptr(), n_lo, n_hi,
                                                   power, np.float32(div).item(), d_dst.data_ptr(), ld_dst, 37, None) != 0
