"""CPU-only checks of the product's host side (no GPU, no compute calls):
 * libsmilehip.so loads and exports every function include/smilehip.h declares;
 * the host-generated tables (window, mel bank, DCT, lifter) and the integer
   geometry equal the oracle's (= the reference's) bit for bit, for the five
   BASELINE geometries;
 * compute entry points refuse to run without a device (no CPU fallback);
 * the reciprocal-based R0 division is exact for all 65536 int16 values.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from opensmile_amd import capi
    capi.load()
    return capi


def test_library_exports_every_declared_symbol(capi):
    hdr = open(os.path.join(ROOT, "include", "smilehip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(smilehip_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = C.CDLL(capi.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, f"declared in smilehip.h but not exported: {missing}"
    # and the Python mirror binds exactly the declared set
    assert set(capi.SYMBOLS) == declared


def test_comm_library_exports_every_declared_symbol():
    """include/smilehip_comm.h <-> libsmilehip_comm.so (the optional RCCL gather; no compute call without a GPU)"""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "smilehip_comm.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(smilehip_comm_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) == 14
    lib = C.CDLL(os.path.join(ROOT, "opensmile_amd", "libsmilehip_comm.so"))
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, f"declared in smilehip_comm.h but not exported: {missing}"


def test_version(capi):
    assert capi.load().smilehip_version() == 0x000100


def _pair(capi, oracle, **kw):
    cfg = capi.mfcc12_0_d_a_config()
    oc = oracle.default_cfg()
    m = {"preemph": "preemph_enable"}
    for k, v in kw.items():
        setattr(cfg, k, v)
        setattr(oc, m.get(k, k), v)
    return cfg, oc


GEOMS = [
    dict(),                                                              # MFCC12_0_D_A @16k
    dict(sample_rate=44100.0),                                           # config 1 (example wav)
    dict(use_power=0, first_mfcc=1, last_mfcc=12),                       # IS09 mel/mfcc settings
    dict(frame_size_sec=0.020, zero_pad_symmetric=1, lofreq=20.0, first_mfcc=1, last_mfcc=14, preemph=0),  # ComParE 20 ms
    dict(frame_size_sec=0.060, win_func=3, zero_pad_symmetric=1, lofreq=20.0, first_mfcc=1, last_mfcc=4, preemph=0),  # 60 ms gauss
]


@pytest.mark.parametrize("kw", GEOMS)
def test_tables_and_geometry_match_oracle(capi, oracle, kw):
    cfg, oc = _pair(capi, oracle, **kw)
    p = capi.Plan(None, cfg)                 # host-only plan
    g, og = p.geometry, oracle.geometry(oc)
    assert (g.frame_size, g.frame_step, g.fft_size, g.n_bins) == (og.N, og.H, og.Nfft, og.K)
    assert g.fft_frame_size_sec == og.frame_size_sec_fft
    win, coef, chan, cos, lif = oracle.export_tables(oc)
    assert np.array_equal(p.window().view(np.uint32), win.view(np.uint32))
    assert np.array_equal(p.mel_weights().view(np.uint32), coef.view(np.uint32))
    assert np.array_equal(p.mel_chanmap(), chan)
    assert np.array_equal(p.lifter().view(np.uint32), lif.view(np.uint32))
    # product stores DCT rows in OUTPUT order (HTK: c1..cN then c0, mfcc.cpp:255-258)
    n = cfg.last_mfcc - cfg.first_mfcc + 1
    rows = p.dct()
    for r in range(n):
        i0 = r
        if cfg.mfcc_htk_compatible and cfg.first_mfcc == 0:
            i0 = 0 if r == n - 1 else r + 1
        assert np.array_equal(rows[r].view(np.uint32), cos[i0].view(np.uint32))
    for S in (0, og.N - 1, og.N, og.N + og.H - 1, og.N + og.H, 160000):
        assert p.num_frames(S) == oracle.lib().lldo_num_frames(S, og.N, og.H)
    p.close()


def test_no_cpu_fallback(capi):
    """Without a HIP device every compute path must fail loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.SmileHipError):
        capi.Context(0)
    p = capi.Plan(None)
    with pytest.raises(capi.SmileHipError):
        capi.Batch(p, np.array([0, 16000], np.int64))
    with pytest.raises(capi.SmileHipError):
        capi.extract_mfcc([np.zeros(16000, np.int16)])


def test_config_validation(capi):
    cfg = capi.mfcc12_0_d_a_config()
    cfg.struct_size = 4
    with pytest.raises(capi.SmileHipError):
        capi.Plan(None, cfg)
    cfg = capi.mfcc12_0_d_a_config()
    cfg.win_func = 99
    with pytest.raises(capi.SmileHipError):
        capi.Plan(None, cfg)
    cfg = capi.mfcc12_0_d_a_config()
    cfg.n_bands = 1000
    with pytest.raises(capi.SmileHipError):
        capi.Plan(None, cfg)


def test_r0_reciprocal_division_exact_for_all_int16():
    """lld_device.hpp pcm16_to_float: q0 = s*r; e = fma(-q0, 32767, s); q = fma(e, r, q0)
    must equal the correctly rounded s / 32767.0f (smileUtil.c:2527-2535)."""
    s = np.arange(-32768, 32768, dtype=np.int32).astype(np.float32)
    ref = s / np.float32(32767.0)
    r = np.float32(1.0) / np.float32(32767.0)
    q0 = s * r
    # emulate fmaf exactly through float64 (products of two float32 are exact in float64)
    e = (-(q0.astype(np.float64)) * 32767.0 + s.astype(np.float64)).astype(np.float32)
    q = (e.astype(np.float64) * np.float64(r) + q0.astype(np.float64)).astype(np.float32)
    assert np.array_equal(q.view(np.uint32), ref.view(np.uint32))


def test_log_table_is_what_its_generator_writes(tmp_path):
    """opensmile_amd/csrc/log_table.inc (the 128-entry table of log_d, lld_device.hpp) is generated, not edited: running
    tools/gen_log_table.py again gives the committed file byte for byte, and every entry satisfies what the kernel relies on
    (|z invc - 1| < 2^-7 over its sub-interval, log c within 2^-63 of the table value; c = 1 for the two intervals at 1)."""
    import math
    import re
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc = os.path.join(root, "opensmile_amd", "csrc", "log_table.inc")
    text = open(inc).read()
    # a private copy of the tree layout the generator writes into
    work = tmp_path / "w"
    (work / "tools").mkdir(parents=True)
    (work / "opensmile_amd" / "csrc").mkdir(parents=True)
    shutil.copy(os.path.join(root, "tools", "gen_log_table.py"), work / "tools" / "gen_log_table.py")
    subprocess.run([sys.executable, str(work / "tools" / "gen_log_table.py")], check=True, capture_output=True)
    assert (work / "opensmile_amd" / "csrc" / "log_table.inc").read_text() == text
    pairs = re.findall(r"\{(-?0x[0-9a-f.]+p[+-]\d+), (-?0x[0-9a-f.]+p[+-]\d+)\}", text)
    assert len(pairs) == 128
    for i, (a, b) in enumerate(pairs):
        invc, logc = float.fromhex(a), float.fromhex(b)
        lo, w = (0.6875 + i * 2.0 ** -8, 2.0 ** -8) if i < 80 else (1.0 + (i - 80) * 2.0 ** -7, 2.0 ** -7)
        assert max(abs(lo * invc - 1.0), abs((lo + w) * invc - 1.0)) < 2.0 ** -7 + 1e-12
        if i in (79, 80):
            assert invc == 1.0 and logc == 0.0
        else:
            assert abs(math.log(1.0 / invc) - logc) < 1e-15
