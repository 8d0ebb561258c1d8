"""ctypes binding of the CPU oracle (oracle/lld_oracle.c) and helpers to run the
real reference binary (oracle/_ref/SMILExtract).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg. The product (opensmile_amd/, include/) never
imports this module.
"""
import ctypes as C
import os
import struct
import subprocess
import tempfile
import wave

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")
LIB_PATH = os.path.join(HERE, "liblld_oracle.so")

WIN = {"rect": 0, "hann": 1, "ham": 2, "gauss": 3, "sine": 4, "tri": 5,
       "bartlett": 6, "lanczos": 7}


class MfccCfg(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_double), ("frame_size_sec", C.c_double),
        ("frame_step_sec", C.c_double),
        ("preemph_enable", C.c_int), ("preemph_k", C.c_float), ("preemph_de", C.c_int),
        ("win_func", C.c_int), ("win_sigma", C.c_double), ("win_gain", C.c_double),
        ("win_offset", C.c_double),
        ("zero_pad_symmetric", C.c_int),
        ("n_bands", C.c_int), ("lofreq", C.c_float), ("hifreq", C.c_float),
        ("use_power", C.c_int), ("mel_htk_compatible", C.c_int),
        ("first_mfcc", C.c_int), ("last_mfcc", C.c_int), ("cep_lifter", C.c_float),
        ("mfcc_htk_compatible", C.c_int), ("melfloor", C.c_float),
        ("n_delta", C.c_int), ("delta_win", C.c_int),
    ]


class Geom(C.Structure):
    _fields_ = [("N", C.c_long), ("H", C.c_long), ("Nfft", C.c_long), ("K", C.c_long),
                ("frame_size_sec_fft", C.c_double)]


def build():
    """(Re)build liblld_oracle.so (and oracle/_ref when /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", HERE, "oracle"], check=True)
    if os.path.isdir("/root/reference/src"):
        subprocess.run(["make", "-s", "-j8", "-C", HERE, "ref"], check=True,
                       stdout=subprocess.DEVNULL)


_lib = None
_ref_dsp = None
_hook_keepalive = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        f32p = C.POINTER(C.c_float)
        L.lldo_default_mfcc12_cfg.argtypes = [C.POINTER(MfccCfg)]
        L.lldo_geometry.argtypes = [C.POINTER(MfccCfg), C.POINTER(Geom)]
        L.lldo_num_frames.restype = C.c_long
        L.lldo_num_frames.argtypes = [C.c_long, C.c_long, C.c_long]
        L.lldo_mfcc_chain.restype = C.c_long
        L.lldo_mfcc_chain.argtypes = [C.POINTER(MfccCfg), C.c_void_p, C.c_long, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lldo_delta_regression.restype = C.c_long
        L.lldo_delta_regression.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_void_p]
        L.lldo_set_rfft_hook.argtypes = [C.c_void_p]
        L.lldo_set_rfft_hook2.argtypes = [C.c_void_p]
        L.lldo_is09_chain.restype = C.c_long
        L.lldo_is09_chain.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
        L.lldo_pcm16_to_float.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
        L.lldo_window_table.argtypes = [C.c_int, C.c_long, C.c_double, C.c_double, C.c_void_p]
        L.lldo_window_table.restype = C.c_int
        L.lldo_rfft_frame.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int]
        _lib = L
    return _lib


def have_ref():
    return os.path.exists(os.path.join(REF_DIR, "SMILExtract"))


def use_reference_fft(enable=True):
    """Plug the reference's own Ooura rdft (oracle/_ref/libref_dsp.so, compiled
    from src/dspcore/fftsg.c) into the restatement -- pins every non-FFT stage
    bit-for-bit against the real binary."""
    global _ref_dsp
    if not enable:
        lib().lldo_set_rfft_hook(None)
        lib().lldo_set_rfft_hook2(None)
        return True
    p = os.path.join(REF_DIR, "libref_dsp.so")
    if not os.path.exists(p):
        return False
    if _ref_dsp is None:
        _ref_dsp = C.CDLL(p)
    lib().lldo_set_rfft_hook(C.cast(_ref_dsp.rdft, C.c_void_p))
    lib().lldo_set_rfft_hook2(C.cast(_ref_dsp.rdft, C.c_void_p))
    return True


def default_cfg():
    c = MfccCfg()
    lib().lldo_default_mfcc12_cfg(C.byref(c))
    return c


def geometry(cfg):
    g = Geom()
    lib().lldo_geometry(C.byref(cfg), C.byref(g))
    return g


def mfcc_chain(cfg, pcm, taps=False):
    """pcm: int16 1-D array. Returns (T x Dtot) float32 [, dict of taps]."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    g = geometry(cfg)
    T = lib().lldo_num_frames(len(pcm), g.N, g.H)
    D = (cfg.last_mfcc - cfg.first_mfcc + 1) * (1 + cfg.n_delta)
    out = np.zeros((max(T, 0), D), dtype=np.float32)
    if T <= 0:
        return (out, {}) if taps else out
    tp = {}
    ptrs = [None] * 4
    if taps:
        tp = {"win": np.zeros((T, g.N), np.float32), "fft": np.zeros((T, g.Nfft), np.float32),
              "mag": np.zeros((T, g.K), np.float32), "mel": np.zeros((T, cfg.n_bands), np.float32)}
        ptrs = [tp[k].ctypes.data for k in ("win", "fft", "mag", "mel")]
    lib().lldo_mfcc_chain(C.byref(cfg), pcm.ctypes.data, len(pcm), out.ctypes.data, *ptrs)
    return (out, tp) if taps else out


def delta_regression(x, W):
    x = np.ascontiguousarray(x, dtype=np.float32)
    T, D = x.shape
    y = np.zeros((T + W, D), np.float32)
    lib().lldo_delta_regression(x.ctypes.data, T, D, W, y.ctypes.data)
    return y


# ------------------------------------------------------------ real reference
def write_wav(path, pcm, fs=16000):
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(fs)
        w.writeframes(np.ascontiguousarray(pcm, dtype="<i2").tobytes())


def read_htk(path):
    """HTK parameter file as written by cHtkSink (src/iocore/htkSink.cpp:93-105,
    183-202): 12-byte big-endian header + big-endian float32 rows."""
    b = open(path, "rb").read()
    n, period, size, kind = struct.unpack(">IIHH", b[:12])
    a = np.frombuffer(b[12:], dtype=">f4").astype(np.float32)
    return a.reshape(n, size // 4), period, kind


def run_reference(conf_rel, pcm, fs=16000, extra_args=()):
    """Run the real SMILExtract (oracle/_ref) on one utterance; returns the HTK
    output matrix. conf_rel is relative to the reference's config/ directory."""
    exe = os.path.join(REF_DIR, "SMILExtract")
    conf = os.path.join(REF_DIR, "config", conf_rel)
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        wav = os.path.join(td, "in.wav")
        out = os.path.join(td, "out.htk")
        write_wav(wav, pcm, fs)
        subprocess.run([exe, "-C", conf, "-I", wav, "-O", out, "-l", "0",
                        *extra_args], check=True, cwd=td,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        if not os.path.exists(out):
            return np.zeros((0, 0), np.float32)
        return read_htk(out)[0]


def delta_chain(x, W, n_orders):
    """Tick-accurate delta chain; returns array (n_orders, T, D)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    T, D = x.shape
    y = np.zeros((n_orders, T, D), np.float32)
    L = lib()
    L.lldo_delta_chain.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int, C.c_void_p]
    L.lldo_delta_chain.restype = None
    L.lldo_delta_chain(x.ctypes.data, T, D, W, n_orders, y.ctypes.data)
    return y


def export_tables(cfg):
    """(window f32[N], mel coef f32[K], chanmap i32[K], costable f32[n_mfcc, n_bands], lifter f32[n_mfcc])"""
    g = geometry(cfg)
    n_mfcc = cfg.last_mfcc - cfg.first_mfcc + 1
    win = np.zeros(g.N, np.float32)
    coef = np.zeros(g.K, np.float32)
    chan = np.zeros(g.K, np.int32)
    cos = np.zeros((n_mfcc, cfg.n_bands), np.float32)
    lif = np.zeros(n_mfcc, np.float32)
    L = lib()
    L.lldo_export_tables.argtypes = [C.POINTER(MfccCfg)] + [C.c_void_p] * 5
    L.lldo_export_tables.restype = None
    L.lldo_export_tables(C.byref(cfg), win.ctypes.data, coef.ctypes.data, chan.ctypes.data,
                         cos.ctypes.data, lif.ctypes.data)
    return win, coef, chan, cos, lif


def is09_chain(pcm, raw=False):
    """IS09_emotion LLD level as the LLD sinks see it: (T+1) x 32 [, T x 16 pre-SMA columns]."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    rows = lib().lldo_is09_chain(pcm.ctypes.data, len(pcm), None, None)
    out = np.zeros((max(rows, 0), 32), np.float32)
    r16 = np.zeros((max(rows - 1, 0), 16), np.float32)
    if rows > 0:
        lib().lldo_is09_chain(pcm.ctypes.data, len(pcm), out.ctypes.data, r16.ctypes.data)
    return (out, r16) if raw else out


def run_reference_lld(conf_rel, pcm, fs=16000):
    """Real SMILExtract, LLD-level HTK output (-lldhtkoutput) of a standard_data_output config."""
    exe = os.path.join(REF_DIR, "SMILExtract")
    conf = os.path.join(REF_DIR, "config", conf_rel)
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        wav = os.path.join(td, "in.wav")
        out = os.path.join(td, "lld.htk")
        write_wav(wav, pcm, fs)
        subprocess.run([exe, "-C", conf, "-I", wav, "-lldhtkoutput", out, "-l", "0"], check=True, cwd=td,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        if not os.path.exists(out):
            return np.zeros((0, 0), np.float32)
        return read_htk(out)[0]


def pcm_convert(raw, n_bps, n_bits, n_chan, mixdown=True):
    """Oracle restatement of smilePcm_convertSamples for every sample format."""
    raw = np.ascontiguousarray(np.frombuffer(bytes(raw), dtype=np.uint8))
    n = len(raw) // (n_bps * n_chan)
    out = np.zeros(n if mixdown else (n, n_chan), np.float32)
    L = lib()
    L.lldo_pcm_convert.restype = C.c_long
    L.lldo_pcm_convert.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, C.c_void_p]
    assert L.lldo_pcm_convert(raw.ctypes.data, n_bps, n_bits, n_chan, int(mixdown), n, out.ctypes.data) == n
    return out


class _WaveParameters(C.Structure):       # sWaveParameters, src/include/smileutil/smileUtil.h:650-661
    _fields_ = [("sampleRate", C.c_long), ("sampleType", C.c_int), ("nChan", C.c_int), ("blockSize", C.c_int),
                ("nBPS", C.c_int), ("nBits", C.c_int), ("byteOrder", C.c_int), ("memOrga", C.c_int),
                ("nBlocks", C.c_long), ("headerOffset", C.c_int)]


def ref_pcm_convert(raw, n_bps, n_bits, n_chan, mixdown=True):
    """The REAL smilePcm_convertSamples (oracle/_ref/libref_dsp.so, compiled from smileUtil.c); None if not built."""
    path = os.path.join(REF_DIR, "libref_dsp.so")
    if not os.path.exists(path):
        return None
    R = C.CDLL(path)
    raw = np.ascontiguousarray(np.frombuffer(bytes(raw), dtype=np.uint8))
    n = len(raw) // (n_bps * n_chan)
    out = np.zeros(n if mixdown else (n, n_chan), np.float32)
    wp = _WaveParameters(16000, 1, n_chan, n_bps * n_chan, n_bps, n_bits, 0, 0, n, 44)
    R.smilePcm_convertSamples.restype = C.c_int
    R.smilePcm_convertSamples.argtypes = [C.c_void_p, C.POINTER(_WaveParameters), C.c_void_p, C.c_int, C.c_int, C.c_int]
    R.smilePcm_convertSamples(raw.ctypes.data, C.byref(wp), out.ctypes.data, 1 if mixdown else n_chan, n, int(mixdown))
    return out


def run_reference_func(conf_rel, pcm, fs=16000):
    """Real SMILExtract, functionals-level HTK output (-htkoutput) of a standard_data_output
    config, together with the LLD level (-lldhtkoutput): (func[1 x F] or empty, lld[rows x D])."""
    exe = os.path.join(REF_DIR, "SMILExtract")
    conf = os.path.join(REF_DIR, "config", conf_rel)
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        wav = os.path.join(td, "in.wav")
        out = os.path.join(td, "func.htk")
        lld = os.path.join(td, "lld.htk")
        write_wav(wav, pcm, fs)
        subprocess.run([exe, "-C", conf, "-I", wav, "-htkoutput", out, "-lldhtkoutput", lld, "-l", "0"], check=True,
                       cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        f = read_htk(out)[0] if os.path.exists(out) else np.zeros((0, 0), np.float32)
        x = read_htk(lld)[0] if os.path.exists(lld) else np.zeros((0, 0), np.float32)
        return f, x


FUNC_BITS = ["max", "min", "range", "maxpos", "minpos", "amean", "maxameandist", "minameandist", "linregc1", "linregc2",
             "linregerrA", "linregerrQ", "variance", "stddev", "skewness", "kurtosis", "amean_m"]
FUNC_IS09 = sum(1 << FUNC_BITS.index(n) for n in ["max", "min", "range", "maxpos", "minpos", "amean", "linregc1", "linregc2",
                                                    "linregerrQ", "stddev", "skewness", "kurtosis"])


def functionals(x, mask=FUNC_IS09):
    """rows x cols matrix -> cols x count(mask) functionals (cFunctionals, frameMode=full)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    L = lib()
    L.lldo_functionals_count.restype = C.c_int
    L.lldo_functionals_count.argtypes = [C.c_uint32]
    L.lldo_functionals.restype = C.c_int
    L.lldo_functionals.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_uint32, C.c_void_p]
    per = L.lldo_functionals_count(mask)
    rows, cols = x.shape
    out = np.zeros((cols, per), np.float32)
    if rows > 0:
        L.lldo_functionals(x.ctypes.data, cols, rows, cols, mask, out.ctypes.data)
    return out


def compare_set_is13(on):
    """Switch the ComParE chain restatements (compare_ab_chain, compare_f0_chain, pitch_jitter, compare_lld_chain) to
    config/is09-13/IS13_ComParE.conf (zeroPadSymmetric = 0, useBrokenJitterThresh = 1); off = ComParE_2016.conf."""
    L = lib()
    L.lldo_compare_set_is13.restype = None
    L.lldo_compare_set_is13.argtypes = [C.c_int]
    L.lldo_compare_set_is13(1 if on else 0)


def is13_func_spec(inst):
    """The six cFunctionals instances of config/is09-13/IS13_ComParE_core.func.conf.inc."""
    s = compare16_func_spec(inst)
    s.mom_ratio_limit = 0
    s.reg_centroid_abs = s.reg_centroid_limit = s.reg_ratio_limit = s.reg_norm_inputs = 0
    s.reg_norm_coeff = 0
    s.pk_ratio_limit = 0
    return s


def compare_b_extra(pcm):
    """Row T60+1 of ComParE's group-B levels: (110,) = 55 sma values + 55 deltas (what [is13_functionalsB] sees beyond
    the rows of the LLD sinks); None if the utterance yields no rows."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    L = lib()
    L.lldo_compare_ab_chain.restype = C.c_long
    L.lldo_compare_ab_chain.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
    L.lldo_compare_set_b_extra.restype = None
    L.lldo_compare_set_b_extra.argtypes = [C.c_void_p]
    rows = L.lldo_compare_ab_chain(pcm.ctypes.data, len(pcm), None, None)
    if rows <= 0:
        return None
    out = np.zeros((rows, 118), np.float32)
    ex = np.zeros(110, np.float32)
    L.lldo_compare_set_b_extra(ex.ctypes.data)
    try:
        L.lldo_compare_ab_chain(pcm.ctypes.data, len(pcm), out.ctypes.data, None)
    finally:
        L.lldo_compare_set_b_extra(None)
    return ex


def plp_chain(pcm):
    """config/plp/PLP_0_D_A.conf: T x 18 [plp c1..c5,c0 | delta | accel]."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    L = lib()
    L.lldo_plp_chain.restype = C.c_long
    L.lldo_plp_chain.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
    T = L.lldo_plp_chain(pcm.ctypes.data, len(pcm), None)
    out = np.zeros((max(T, 0), 18), np.float32)
    if T > 0:
        L.lldo_plp_chain(pcm.ctypes.data, len(pcm), out.ctypes.data)
    return out


def compare_ab_chain(pcm, raw=False):
    """ComParE_2016 LLD groups A+B as the LLD sinks see them: rows x 118 [, rows-1 x 59 pre-SMA]."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    L = lib()
    L.lldo_compare_ab_chain.restype = C.c_long
    L.lldo_compare_ab_chain.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
    rows = L.lldo_compare_ab_chain(pcm.ctypes.data, len(pcm), None, None)
    out = np.zeros((max(rows, 0), 118), np.float32)
    r59 = np.zeros((max(rows - 1, 0), 59), np.float32)
    if rows > 0:
        L.lldo_compare_ab_chain(pcm.ctypes.data, len(pcm), out.ctypes.data, r59.ctypes.data)
    return (out, r59) if raw else out


F0_TAPS_CONF = os.path.join(HERE, "conf", "compare_f0_taps.conf")


def run_reference_taps(pcm, names=("hps", "shs", "vit", "pitch", "e60", "jit", "nzsmo", "nzsmo_de"), fs=16000,
                       conf=None):
    """Real SMILExtract on ComParE_2016 with HTK taps on the F0-group levels (oracle/conf/compare_f0_taps.conf).
    Returns {tap name: matrix} plus "lld" (the 130-column LLD level)."""
    exe = os.path.join(REF_DIR, "SMILExtract")
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        wav = os.path.join(td, "in.wav")
        write_wav(wav, pcm, fs)
        subprocess.run([exe, "-C", conf or F0_TAPS_CONF, "-I", wav, "-lldhtkoutput", "lld.htk", "-l", "0"], check=True,
                       cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = {}
        for n in tuple(names) + ("lld",):
            p = os.path.join(td, "lld.htk" if n == "lld" else "tap_%s.htk" % n)
            out[n] = read_htk(p)[0] if os.path.exists(p) else np.zeros((0, 0), np.float32)
        return out


FUNC_TAPS_CONF = os.path.join(HERE, "conf", "compare_func_taps.conf")
FUNC_TAPS = ("a_smo", "a_de", "b_smo", "b_de", "nz_smo", "nz_de", "f0_smo")


def run_reference_func_taps(pcm, fs=16000, is13=False):
    """Real SMILExtract on ComParE_2016 (oracle/conf/compare_func_taps.conf): {"func": (6373,) or empty, "names": list,
    "lld": LLD level, tap name: the level a cFunctionals instance reads, all its rows}."""
    exe = os.path.join(REF_DIR, "SMILExtract")
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        wav = os.path.join(td, "in.wav")
        write_wav(wav, pcm, fs)
        conf = FUNC_TAPS_CONF
        if is13:                                  # the same taps on IS13_ComParE.conf (identical level names)
            conf = os.path.join(td, "taps_is13.conf")
            with open(conf, "w") as f:
                f.write(open(FUNC_TAPS_CONF).read().replace("../_ref/config/compare16/ComParE_2016.conf",
                                                            os.path.join(REF_DIR, "config", "is09-13", "IS13_ComParE.conf")))
        subprocess.run([exe, "-C", conf, "-I", wav, "-htkoutput", "func.htk", "-csvoutput", "func.csv",
                        "-lldhtkoutput", "lld.htk", "-l", "0"], check=True, cwd=td, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        out = {}
        for n in FUNC_TAPS + ("lld", "func"):
            p = os.path.join(td, ("%s.htk" % n) if n in ("lld", "func") else "tap_%s.htk" % n)
            out[n] = read_htk(p)[0] if os.path.exists(p) else np.zeros((0, 0), np.float32)
        out["func"] = out["func"][0] if out["func"].shape[0] else np.zeros(0, np.float32)
        csv = os.path.join(td, "func.csv")
        out["names"] = open(csv).readline().strip().split(";")[2:] if os.path.exists(csv) else []
        return out


def set_sample_rate(rate):
    """Sample rate of the input of the chain functions (lld_oracle.c: g_lldo_rate; 16000 unless set). Reset it after use."""
    L = lib()
    L.lldo_set_sample_rate.argtypes = [C.c_double]
    L.lldo_set_sample_rate.restype = None
    L.lldo_set_sample_rate(float(rate))


def get_sample_rate():
    L = lib()
    L.lldo_get_sample_rate.restype = C.c_double
    return L.lldo_get_sample_rate()


def bins_of(frame_sec):
    """Number of magnitude bins of a frame of frame_sec seconds at the oracle's current rate (FFT = next power of two)."""
    n = int(round(frame_sec * get_sample_rate()))
    nfft = 1
    while nfft < n:
        nfft *= 2
    return nfft // 2 + 1


def compare_f0_chain(pcm, taps=False):
    """ComParE_2016 F0 group, level is13_pitchG60: T60 x 2 [F0final, voicingFinalUnclipped];
    taps=True also returns {hps, shs, vit, e60} (the levels of the same names, see lld_oracle_f0.c)."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    L = lib()
    L.lldo_compare_f0_chain.restype = C.c_long
    L.lldo_compare_f0_chain.argtypes = [C.c_void_p, C.c_long] + [C.c_void_p] * 5
    T = L.lldo_compare_f0_chain(pcm.ctypes.data, len(pcm), None, None, None, None, None)
    T = max(T, 0)
    out = np.zeros((T, 2), np.float32)
    t = {"hps": np.zeros((T, bins_of(0.060)), np.float32), "shs": np.zeros((T, 21), np.float32),
         "vit": np.zeros((T, 2), np.float32), "e60": np.zeros((T, 1), np.float32)}
    if T > 0:
        r = L.lldo_compare_f0_chain(pcm.ctypes.data, len(pcm), out.ctypes.data, t["hps"].ctypes.data,
                                    t["shs"].ctypes.data, t["vit"].ctypes.data, t["e60"].ctypes.data)
        if r < 0:
            raise RuntimeError("lldo_compare_f0_chain failed")
    return (out, t) if taps else out


def pitch_jitter(pcm, f0, N=960, H=160, fs=16000.0, step=0.010):
    """cPitchJitter as [is13_pitchJitter] configures it: (T x 4) [jitterLocal, jitterDDP, shimmerLocal, logHNR]
    from the utterance and its F0final contour (one value per 60 ms frame)."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    L = lib()
    x = np.zeros(len(pcm), np.float32)
    L.lldo_pcm16_to_float.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
    L.lldo_pcm16_to_float.restype = None
    L.lldo_pcm16_to_float(pcm.ctypes.data, len(pcm), x.ctypes.data)
    f0 = np.ascontiguousarray(f0, dtype=np.float32)
    out = np.zeros((len(f0), 4), np.float32)
    L.lldo_pitch_jitter.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_double, C.c_double, C.c_void_p]
    L.lldo_pitch_jitter.restype = None
    L.lldo_pitch_jitter(x.ctypes.data, len(x), f0.ctypes.data, len(f0), N, H, fs, step, out.ctypes.data)
    return out


def compare_lld_chain(pcm):
    """ComParE_2016's whole LLD level as its LLD sinks see it: (T60+1) x 130."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    L = lib()
    L.lldo_compare_lld_chain.restype = C.c_long
    L.lldo_compare_lld_chain.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
    rows = L.lldo_compare_lld_chain(pcm.ctypes.data, len(pcm), None)
    out = np.zeros((max(rows, 0), 130), np.float32)
    if rows > 0 and L.lldo_compare_lld_chain(pcm.ctypes.data, len(pcm), out.ctypes.data) != rows:
        raise RuntimeError("lldo_compare_lld_chain failed")
    return out


def compare_f0_lld(pcm):
    """The F0 group's 12 LLD columns [6 smoothed | 6 deltas], T60+1 rows."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    L = lib()
    L.lldo_compare_f0_lld.restype = C.c_long
    L.lldo_compare_f0_lld.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
    rows = L.lldo_compare_f0_lld(pcm.ctypes.data, len(pcm), None)
    out = np.zeros((max(rows, 0), 12), np.float32)
    if rows > 0:
        L.lldo_compare_f0_lld(pcm.ctypes.data, len(pcm), out.ctypes.data)
    return out


HTK_VARIANTS = {  # name: (config file, plp, energy, cms)
    "MFCC12_0_D_A": ("mfcc/MFCC12_0_D_A.conf", 0, 0, 0), "MFCC12_E_D_A": ("mfcc/MFCC12_E_D_A.conf", 0, 1, 0),
    "MFCC12_0_D_A_Z": ("mfcc/MFCC12_0_D_A_Z.conf", 0, 0, 1), "MFCC12_E_D_A_Z": ("mfcc/MFCC12_E_D_A_Z.conf", 0, 1, 1),
    "PLP_0_D_A": ("plp/PLP_0_D_A.conf", 1, 0, 0), "PLP_E_D_A": ("plp/PLP_E_D_A.conf", 1, 1, 0),
    "PLP_0_D_A_Z": ("plp/PLP_0_D_A_Z.conf", 1, 0, 1), "PLP_E_D_A_Z": ("plp/PLP_E_D_A_Z.conf", 1, 1, 1),
}


def htk_variant_chain(name, pcm):
    """The eight configs of config/mfcc and config/plp: T x 3*(cepstra [+ energy])."""
    _, plp, energy, cms = HTK_VARIANTS[name]
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    L = lib()
    L.lldo_htk_variant_chain.restype = C.c_long
    L.lldo_htk_variant_chain.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p]
    T = max(L.lldo_htk_variant_chain(plp, energy, cms, pcm.ctypes.data, len(pcm), None), 0)
    D = ((5 if plp else 12) + (0 if energy else 1) + (1 if energy else 0)) * 3
    out = np.zeros((T, D), np.float32)
    if T > 0:
        L.lldo_htk_variant_chain(plp, energy, cms, pcm.ctypes.data, len(pcm), out.ctypes.data)
    return out


# ---- general functionals engine (lld_oracle_funcspec.c) ---------------------------------------------------------
FAM = {"Extremes": 0, "Means": 1, "Moments": 2, "Regression": 3, "Percentiles": 4, "Times": 5, "Segments": 6, "Lpc": 7,
       "Peaks2": 8, "Onset": 9, "Peaks": 10, "Crossings": 11, "DCT": 12, "Samples": 13, "Modulation": 14}
NORM = {"segment": 0, "second": 1, "frame": 2}
EXT_NAMES = ["max", "min", "range", "maxPos", "minPos", "amean", "maxameandist", "minameandist"]
MEANS_NAMES = ["amean", "absmean", "qmean", "nzamean", "nzabsmean", "nzqmean", "nzgmean", "nnz", "flatness", "posamean",
               "negamean", "posqmean", "posrqmean", "negqmean", "negrqmean", "rqmean", "nzrqmean"]
MOM_NAMES = ["variance", "stddev", "skewness", "kurtosis", "amean", "stddevNorm"]
REG_NAMES = ["linregc1", "linregc2", "linregerrA", "linregerrQ", "qregc1", "qregc2", "qregc3", "qregerrA", "qregerrQ",
             "centroid", "qregls", "qregrs", "qregx0", "qregy0", "qregyr", "qregy0nn", "qregc3nn", "qregyrnn"]
PCT_NAMES = ["quartile1", "quartile2", "quartile3", "iqr1-2", "iqr2-3", "iqr1-3"]
TIMES_NAMES = ["upleveltime25", "downleveltime25", "upleveltime50", "downleveltime50", "upleveltime75", "downleveltime75",
               "upleveltime90", "downleveltime90", "risetime", "falltime", "leftctime", "rightctime", "duration"]
PKO_NAMES = ["numPeaks", "meanPeakDist", "peakMean", "peakMeanMeanDist", "peakDistStddev"]
ONS_NAMES = ["onsetPos", "offsetPos", "numOnsets", "numOffsets", "onsetRate"]
SEG_NAMES = ["numSegments", "meanSegLen", "maxSegLen", "minSegLen", "segLenStddev"]
PK_NAMES = ["numPeaks", "meanPeakDist", "meanPeakDistDelta", "peakDistStddev", "peakRangeAbs", "peakRangeRel", "peakMeanAbs",
            "peakMeanMeanDist", "peakMeanRel", "ptpAmpMeanAbs", "ptpAmpMeanRel", "ptpAmpStddevAbs", "ptpAmpStddevRel",
            "minRangeAbs", "minRangeRel", "minMeanAbs", "minMeanMeanDist", "minMeanRel", "mtmAmpMeanAbs", "mtmAmpMeanRel",
            "mtmAmpStddevAbs", "mtmAmpStddevRel", "meanRisingSlope", "maxRisingSlope", "minRisingSlope", "stddevRisingSlope",
            "meanFallingSlope", "maxFallingSlope", "minFallingSlope", "stddevFallingSlope", "covFallingSlope", "covRisingSlope"]


def _mask(names, on):
    return sum(1 << names.index(n) for n in on)


class FuncSpec(C.Structure):
    """lldo_func_spec (oracle/lld_oracle_funcspec.h), field for field."""
    _fields_ = [
        ("n_fam", C.c_int32), ("fam", C.c_int32 * 12), ("non_zero_functs", C.c_int32), ("reserved0", C.c_int32),
        ("period", C.c_double),
        ("ext_mask", C.c_uint32), ("ext_norm", C.c_int32),
        ("means_mask", C.c_uint32), ("means_norm", C.c_int32),
        ("mom_mask", C.c_uint32), ("mom_stddev_norm", C.c_int32), ("mom_ratio_limit", C.c_int32), ("reserved1", C.c_int32),
        ("reg_mask", C.c_uint32), ("reg_centroid_norm", C.c_int32), ("reg_norm_coeff", C.c_int32),
        ("reg_norm_inputs", C.c_int32), ("reg_centroid_abs", C.c_int32), ("reg_centroid_limit", C.c_int32),
        ("reg_ratio_limit", C.c_int32), ("reg_old_buggy_qerr", C.c_int32),
        ("pct_mask", C.c_uint32), ("pct_interp", C.c_int32), ("n_pctl", C.c_int32), ("n_range", C.c_int32),
        ("pctl", C.c_double * 8), ("range_a", C.c_int32 * 8), ("range_b", C.c_int32 * 8),
        ("times_mask", C.c_uint32), ("times_norm", C.c_int32), ("times_buggy_sec_norm", C.c_int32), ("reserved2", C.c_int32),
        ("seg_mask", C.c_uint32), ("seg_norm", C.c_int32), ("seg_algo", C.c_int32), ("seg_max_num", C.c_int32),
        ("seg_min_lng", C.c_int32), ("seg_auto_min_lng", C.c_int32), ("seg_pause_min_lng", C.c_int32),
        ("seg_x_is_rel", C.c_int32), ("seg_n_thresholds", C.c_int32), ("seg_ravg_lng", C.c_int32),
        ("seg_x", C.c_float), ("seg_thresholds", C.c_float * 8), ("seg_range_rel_threshold", C.c_float),
        ("lpc_gain", C.c_int32), ("lpc_coeffs", C.c_int32), ("lpc_first", C.c_int32), ("lpc_order", C.c_int32),
        ("pk_mask", C.c_uint32), ("pk_norm", C.c_int32), ("pk_ratio_limit", C.c_int32), ("pk_dyn_rel", C.c_int32),
        ("pk_use_abs", C.c_int32), ("reserved5", C.c_int32),
        ("pk_rel_thresh", C.c_float), ("pk_abs_thresh", C.c_float),
        ("ons_mask", C.c_uint32), ("ons_norm", C.c_int32), ("ons_use_abs", C.c_int32), ("reserved6", C.c_int32),
        ("ons_thr_on", C.c_float), ("ons_thr_off", C.c_float),
        ("pko_mask", C.c_uint32), ("pko_norm", C.c_int32),
        ("crs_mask", C.c_uint32), ("dct_first", C.c_int32), ("dct_last", C.c_int32), ("n_samples", C.c_int32),
        ("sample_pos", C.c_double * 8),
        ("n_quot", C.c_int32), ("quot_a", C.c_int32 * 8), ("quot_b", C.c_int32 * 8),
        ("n_ul", C.c_int32), ("n_dl", C.c_int32), ("reserved7", C.c_int32),
        ("ul", C.c_double * 8), ("dl", C.c_double * 8),
        ("mod_win_frames", C.c_int32), ("mod_step_frames", C.c_int32), ("mod_n_bins", C.c_int32), ("mod_win_func", C.c_int32),
        ("mod_remove_nz_mean", C.c_int32), ("reserved8", C.c_int32),
        ("mod_min_freq", C.c_double), ("mod_max_freq", C.c_double),
    ]


def _spec_common(s, fams, period=0.01):
    s.n_fam = len(fams)
    for i, f in enumerate(fams):
        s.fam[i] = FAM[f]
    s.period = period
    return s


def _set_ext(s):
    s.ext_mask = _mask(EXT_NAMES, ["range", "maxPos", "minPos"])
    s.ext_norm = NORM["segment"]


def _set_pct(s):
    s.pct_mask = 0x3f
    s.pct_interp = 1
    s.n_pctl = 2
    s.pctl[0], s.pctl[1] = 0.01, 0.99
    s.n_range = 1
    s.range_a[0], s.range_b[0] = 0, 1


def _set_mom(s):
    s.mom_mask = _mask(MOM_NAMES, ["stddev", "skewness", "kurtosis"])
    s.mom_ratio_limit = 1


def _set_times(s):
    s.times_mask = _mask(TIMES_NAMES, ["upleveltime25", "upleveltime50", "upleveltime75", "upleveltime90", "risetime",
                                       "leftctime"])
    s.times_norm = NORM["segment"]


def _set_lpc(s):
    s.lpc_gain, s.lpc_coeffs, s.lpc_first, s.lpc_order = 1, 1, 0, 5


def _set_reg(s, norm_coeff):
    s.reg_mask = _mask(REG_NAMES, ["linregc1", "linregc2", "linregerrQ", "qregc1", "qregc2", "qregc3", "qregerrQ", "centroid"])
    s.reg_centroid_norm = NORM["segment"]
    s.reg_norm_coeff = norm_coeff
    s.reg_norm_inputs = s.reg_centroid_abs = s.reg_centroid_limit = s.reg_ratio_limit = 1


def _set_pk(s):
    s.pk_mask = _mask(PK_NAMES, ["meanPeakDist", "peakDistStddev", "peakRangeAbs", "peakRangeRel", "peakMeanAbs",
                                 "peakMeanMeanDist", "peakMeanRel", "minRangeRel", "meanRisingSlope", "stddevRisingSlope",
                                 "meanFallingSlope", "stddevFallingSlope"])
    s.pk_norm = NORM["second"]
    s.pk_ratio_limit = 1
    s.pk_rel_thresh = 0.1


def compare16_func_spec(inst):
    """The six cFunctionals instances of config/compare16/ComParE_2016_core.func.conf.inc: 'A' (= 'B'), 'F0', 'Nz',
    'LLD', 'Delta'."""
    s = FuncSpec()
    if inst in ("A", "B"):
        _spec_common(s, ["Extremes", "Percentiles", "Moments", "Segments", "Times", "Lpc"])
        _set_ext(s); _set_pct(s); _set_mom(s); _set_times(s); _set_lpc(s)
        s.seg_mask = _mask(SEG_NAMES, ["meanSegLen", "maxSegLen", "minSegLen", "segLenStddev"])
        s.seg_norm, s.seg_algo, s.seg_max_num = NORM["second"], 0, 100
        s.seg_min_lng, s.seg_auto_min_lng, s.seg_pause_min_lng = 3, 1, 2
        s.seg_n_thresholds = 2
        s.seg_thresholds[0], s.seg_thresholds[1] = 0.25, 0.75
    elif inst == "F0":
        _spec_common(s, ["Means", "Segments"])
        s.means_mask, s.means_norm = _mask(MEANS_NAMES, ["nnz"]), NORM["segment"]
        s.seg_mask = _mask(SEG_NAMES, ["meanSegLen", "maxSegLen", "minSegLen", "segLenStddev"])
        s.seg_norm, s.seg_algo, s.seg_max_num = NORM["second"], 1, 100
        s.seg_min_lng, s.seg_auto_min_lng, s.seg_pause_min_lng = 3, 1, 2
        s.seg_x = 0.0
    elif inst == "Nz":
        _spec_common(s, ["Means", "Extremes", "Regression", "Percentiles", "Moments", "Times", "Lpc"])
        s.non_zero_functs = 1
        s.means_mask, s.means_norm = _mask(MEANS_NAMES, ["amean", "posamean", "rqmean", "flatness"]), NORM["frame"]
        _set_ext(s); _set_reg(s, 0); _set_pct(s); _set_mom(s); _set_times(s); _set_lpc(s)
    elif inst == "LLD":
        _spec_common(s, ["Means", "Peaks2", "Regression"])
        s.means_mask, s.means_norm = _mask(MEANS_NAMES, ["amean", "rqmean", "flatness"]), NORM["frame"]
        _set_pk(s); _set_reg(s, 2)
    elif inst == "Delta":
        _spec_common(s, ["Means", "Peaks2"])
        s.means_mask, s.means_norm = _mask(MEANS_NAMES, ["posamean", "rqmean", "flatness"]), NORM["frame"]
        _set_pk(s)
    else:
        raise ValueError(inst)
    return s


def funcspec_names(s):
    """Value names of a spec in output order, as cFunctionals::setupNamesForElement builds the suffixes."""
    out = []
    inv = {v: k for k, v in FAM.items()}
    for i in range(s.n_fam):
        f = inv[s.fam[i]]
        if f == "Extremes":
            out += [n for k, n in enumerate(EXT_NAMES) if s.ext_mask >> k & 1]
        elif f == "Means":
            out += [n for k, n in enumerate(MEANS_NAMES) if s.means_mask >> k & 1]
        elif f == "Moments":
            out += [n for k, n in enumerate(MOM_NAMES) if s.mom_mask >> k & 1]
        elif f == "Regression":
            out += [n for k, n in enumerate(REG_NAMES) if s.reg_mask >> k & 1]
        elif f == "Percentiles":
            out += [n for k, n in enumerate(PCT_NAMES) if s.pct_mask >> k & 1]
            out += ["percentile%.1f" % (s.pctl[k] * 100.0) for k in range(s.n_pctl)]
            out += ["pctlrange%d-%d" % (s.range_a[k], s.range_b[k]) for k in range(s.n_range)]
            out += ["pctlquotient%d-%d" % (s.quot_a[k], s.quot_b[k]) for k in range(s.n_quot if s.n_pctl > 0 else 0)]
        elif f == "Times":
            out += [n for k, n in enumerate(TIMES_NAMES) if s.times_mask >> k & 1]
            out += ["upleveltime%.1f" % (s.ul[k] * 100.0) for k in range(s.n_ul)]
            out += ["downleveltime%.1f" % (s.dl[k] * 100.0) for k in range(s.n_dl)]
        elif f == "Segments":
            out += [n for k, n in enumerate(SEG_NAMES) if s.seg_mask >> k & 1]
        elif f == "Lpc":
            out += (["lpgain"] if s.lpc_gain else []) + (["lpc%d" % k for k in range(s.lpc_first, s.lpc_order)]
                                                         if s.lpc_coeffs else [])
        elif f == "Peaks2":
            out += [n for k, n in enumerate(PK_NAMES) if s.pk_mask >> k & 1]
        elif f == "Onset":
            out += [n for k, n in enumerate(ONS_NAMES) if s.ons_mask >> k & 1]
        elif f == "Peaks":
            out += [n for k, n in enumerate(PKO_NAMES) if s.pko_mask >> k & 1]
        elif f == "Crossings":
            out += [n for k, n in enumerate(["zcr", "mcr", "amean"]) if s.crs_mask >> k & 1]
        elif f == "DCT":
            out += ["dct%d" % k for k in range(s.dct_first, s.dct_last + 1)]
        elif f == "Samples":
            out += ["sample%.3f" % s.sample_pos[k] for k in range(s.n_samples)]
        elif f == "Modulation":
            out += ["ModulationSpec%d" % k for k in range(s.mod_n_bins)]
    return out



class ModSpecCfg(C.Structure):
    """lldo_modspec_cfg (oracle/lld_oracle_modspec.h)"""
    _fields_ = [("period", C.c_double), ("min_freq", C.c_double), ("max_freq", C.c_double), ("win_frames", C.c_int32),
                ("step_frames", C.c_int32), ("n_bins", C.c_int32), ("win_func", C.c_int32), ("remove_nz_mean", C.c_int32),
                ("reserved", C.c_int32)]


def modspec_config(period=0.01, win_sec=4.0, step_sec=0.0, win_frames=None, step_frames=None, num_bins=None, resolution=0.5,
                   min_freq=0.5, max_freq=20.0, win_func=2, remove_nz_mean=0):
    """cFunctionalModulation's options (defaults as registered, functionalModulation.cpp:68-82) -> ModSpecCfg"""
    c = ModSpecCfg()
    L = lib()
    L.lldo_modspec_config.restype = None
    L.lldo_modspec_config.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
    L.lldo_modspec_config(C.byref(c), period, win_sec, step_sec, int(win_frames is not None), int(win_frames or 0),
                          int(step_frames is not None), int(step_frames or 0), int(num_bins is not None), int(num_bins or 0),
                          resolution, min_freq, max_freq, win_func, remove_nz_mean)
    return c


def modspec(x, cfg):
    """rows x cols matrix -> cols x n_bins modulation spectra (cFunctionalModulation per column)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    L = lib()
    L.lldo_modspec_apply.restype = C.c_int
    L.lldo_modspec_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
    out = np.zeros((x.shape[1], cfg.n_bins), np.float32)
    for c in range(x.shape[1]):
        col = np.ascontiguousarray(x[:, c])
        if not L.lldo_modspec_apply(C.byref(cfg), col.ctypes.data, col.shape[0], out[c].ctypes.data):
            raise ValueError("modulation spectrum: input outside what the restatement covers")
    return out


def funcspec(x, spec):
    """rows x cols matrix -> cols x count(spec) functionals."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    L = lib()
    L.lldo_funcspec_count.restype = C.c_int
    L.lldo_funcspec_count.argtypes = [C.c_void_p]
    L.lldo_funcspec_apply.restype = C.c_int
    L.lldo_funcspec_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p]
    per = L.lldo_funcspec_count(C.byref(spec))
    if per < 0:
        raise ValueError("unusable functionals spec")
    rows, cols = x.shape
    out = np.zeros((cols, per), np.float32)
    if rows > 0:
        L.lldo_funcspec_apply(C.byref(spec), x.ctypes.data, x.strides[0] // 4, rows, cols, out.ctypes.data)
    return out


# ---------------------------------------------------------------- eGeMAPSv02 (oracle/lld_oracle_gemaps.c)
EGEMAPS_TAPS_CONF = os.path.join(HERE, "conf", "egemaps_taps.conf")
# key -> level of config/egemaps/v02/eGeMAPSv02.conf tapped by oracle/conf/egemaps_taps.conf
EGEMAPS_LEVELS = {
    "loudness": "gemapsv01b_loudness", "lspec": "gemapsv01b_logSpectral", "flux": "egemapsv02_logSpectral_flux",
    "mfcc": "egemapsv02_mfcc", "energy2": "egemapsv02_energyRMS", "formants": "gemapsv01b_formants",
    "pitch": "gemapsv01b_logPitch", "jitter": "gemapsv01b_jitterShimmer", "harm": "gemapsv01b_harmonics",
    "shs": "gemapsv01b_pitchShsG60", "e60": "gemapsv01b_e60", "lpc": "gemapsv01b_lpc",
    "E": "egemapsv02_lldsetE_smo", "F": "egemapsv02_lldsetF_smo", "logf0": "gemapsv01b_lld_single_logF0_smo",
    "loud": "gemapsv01b_loudness_smo", "NoZ": "egemapsv02_lldSetNoF0AndLoudnessZ_smo",
    "NoNz": "egemapsv02_lldSetNoF0AndLoudnessNz_smo", "specV": "egemapsv02_lldSetSpectralNz_smo",
    "specU": "egemapsv02_lldSetSpectralZ_smo", "mag60": "gemapsv01b_fftmagG60", "mag20": "gemapsv01b_fftmagH25",
}
_EG_LV = [("loudness", 1, 20), ("lspec", 4, 20), ("flux", 1, 20), ("mfcc", 4, 20), ("energy2", 1, 20), ("formants", 10, 20),
          ("pitch", 3, 60), ("jitter", 2, 60), ("harm", 6, 60), ("shs", 21, 60), ("e60", 1, 60)]
_EG_SMO = [("E", 10, 20), ("F", 15, 60), ("logf0", 1, 60), ("loud", 1, 20), ("NoZ", 5, 20), ("NoNz", 14, 60), ("specV", 9, 60),
           ("specU", 5, 60)]


class _EgLv(C.Structure):
    _fields_ = [("T20", C.c_long), ("T60", C.c_long), ("P", C.c_long)] + [(k, C.POINTER(C.c_float)) for k, _, _ in _EG_LV]


class _EgSmo(C.Structure):
    _fields_ = [("T20", C.c_long), ("T60", C.c_long), ("P", C.c_long)] + [(k, C.POINTER(C.c_float)) for k, _, _ in _EG_SMO]


def _eg_bind():
    L = lib()
    L.lldo_egemaps_levels.restype = C.c_long
    L.lldo_egemaps_levels.argtypes = [C.c_void_p, C.c_long, C.POINTER(_EgLv)]
    L.lldo_egemaps_smooth.argtypes = [C.POINTER(_EgLv), C.POINTER(_EgSmo)]
    L.lldo_egemaps_lld_chain.restype = C.c_long
    L.lldo_egemaps_lld_chain.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
    L.lldo_egemaps_func.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
    L.lldo_egemaps_func_from_levels.argtypes = [C.POINTER(_EgLv), C.POINTER(_EgSmo), C.c_void_p]
    L.lldo_funcspec_egemaps.argtypes = [C.c_char_p, C.c_void_p]
    return L


def _eg_arr(p, n, m):
    return np.ctypeslib.as_array(p, (n, m)).copy() if n > 0 else np.zeros((0, m), np.float32)


def egemaps_levels(pcm):
    """Every per-frame and every smoothed level of eGeMAPSv02's LLD graph for one utterance:
    {'T20', 'T60', 'P', per-frame levels ..., smoothed levels ...} (keys of EGEMAPS_LEVELS)."""
    L = _eg_bind()
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    lv, sm = _EgLv(), _EgSmo()
    L.lldo_egemaps_levels(pcm.ctypes.data, len(pcm), C.byref(lv))
    out = {"T20": lv.T20, "T60": lv.T60, "P": lv.P}
    for k, m, f in _EG_LV:
        out[k] = _eg_arr(getattr(lv, k), lv.T20 if f == 20 else lv.T60, m)
    if lv.T60 >= 1:
        L.lldo_egemaps_smooth(C.byref(lv), C.byref(sm))
        for k, m, f in _EG_SMO:
            out[k] = _eg_arr(getattr(sm, k), (lv.T20 if f == 20 else lv.T60) + 1, m)
        L.lldo_egemaps_smo_free(C.byref(sm))
    L.lldo_egemaps_levels_free(C.byref(lv))
    return out


def egemaps_lld_chain(pcm):
    """The 25-column LLD level of eGeMAPSv02.conf, T60 + 1 rows."""
    L = _eg_bind()
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    rows = L.lldo_egemaps_lld_chain(pcm.ctypes.data, len(pcm), None)
    out = np.zeros((max(rows, 0), 25), np.float32)
    if rows > 0:
        L.lldo_egemaps_lld_chain(pcm.ctypes.data, len(pcm), out.ctypes.data)
    return out


def egemaps_func(pcm):
    """The 88 functionals of eGeMAPSv02.conf ((0, 88) when the reference writes no vector)."""
    L = _eg_bind()
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    out = np.zeros((1, 88), np.float32)
    r = L.lldo_egemaps_func(pcm.ctypes.data, len(pcm), out.ctypes.data)
    return out if r else np.zeros((0, 88), np.float32)


def egemaps_func_spec(inst):
    s = FuncSpec()
    if not _eg_bind().lldo_funcspec_egemaps(inst.encode(), C.byref(s)):
        raise ValueError(inst)
    return s


def run_reference_egemaps(pcm, fs=16000, levels=()):
    """Real SMILExtract on the unmodified eGeMAPSv02.conf (+ HTK taps, oracle/conf/egemaps_taps.conf): returns
    {'lld': T60+1 x 25, 'func': 1 x 88 (or 0 x 88), requested level keys of EGEMAPS_LEVELS ...}."""
    exe = os.path.join(REF_DIR, "SMILExtract")
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        wav = os.path.join(td, "in.wav")
        write_wav(wav, pcm, fs)
        subprocess.run([exe, "-C", EGEMAPS_TAPS_CONF, "-I", wav, "-lldhtkoutput", "lld.htk", "-htkoutput", "func.htk", "-l", "0"],
                       check=True, cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = {}
        for k, fn, w in (("lld", "lld.htk", 25), ("func", "func.htk", 88)):
            p = os.path.join(td, fn)
            out[k] = read_htk(p)[0] if os.path.exists(p) else np.zeros((0, w), np.float32)
            if out[k].size == 0:
                out[k] = np.zeros((0, w), np.float32)
        for k in levels:
            p = os.path.join(td, "tap_%s.htk" % EGEMAPS_LEVELS[k])
            out[k] = read_htk(p)[0] if os.path.exists(p) else np.zeros((0, 0), np.float32)
        return out


# stage-level entry points of the eGeMAPS restatement (identical-input checks of the per-component HIP operators)
class _GSpec(C.Structure):
    _fields_ = [("K", C.c_long), ("lo", C.c_long), ("hi", C.c_long), ("frq", C.c_void_p), ("prev", C.c_void_p),
                ("have_prev", C.c_int), ("spec_floor", C.c_float), ("log_spec_floor", C.c_float)]


class _SpecRes(C.Structure):
    _fields_ = [("K", C.c_long), ("I", C.c_long), ("kMax", C.c_long), ("target_fs", C.c_double), ("costable", C.c_void_p),
                ("sintable", C.c_void_p)]


def egemaps_spectral_rows(mag, frame_size_sec=512 / 16000.0):
    """cSpectral with the GeMAPS option sets over the frames of one stream: n x K magnitudes -> n x 5."""
    L = lib()
    L.lldo_gspec_init.argtypes = [C.POINTER(_GSpec), C.c_long, C.c_double]
    L.lldo_gspec_frame.argtypes = [C.POINTER(_GSpec), C.c_void_p, C.c_void_p]
    L.lldo_gspec_free.argtypes = [C.POINTER(_GSpec)]
    mag = np.ascontiguousarray(mag, dtype=np.float32)
    out = np.zeros((mag.shape[0], 5), np.float32)
    s = _GSpec()
    L.lldo_gspec_init(C.byref(s), mag.shape[1], frame_size_sec)
    for i in range(mag.shape[0]):
        L.lldo_gspec_frame(C.byref(s), mag[i].ctypes.data, out[i].ctypes.data)
    L.lldo_gspec_free(C.byref(s))
    return out


def egemaps_specresample_rows(spec):
    """cSpecResample of [gemapsv01b_resampLpc]: n x 512 packed spectra of 20 ms frames at 16 kHz -> n x 220."""
    L = lib()
    L.lldo_specresample_init.argtypes = [C.POINTER(_SpecRes), C.c_long, C.c_double, C.c_double, C.c_double, C.c_double]
    L.lldo_specresample_frame.argtypes = [C.POINTER(_SpecRes), C.c_void_p, C.c_void_p]
    L.lldo_specresample_free.argtypes = [C.POINTER(_SpecRes)]
    spec = np.ascontiguousarray(spec, dtype=np.float32)
    r = _SpecRes()
    L.lldo_specresample_init(C.byref(r), spec.shape[1], spec.shape[1] / 16000.0, 320 / 16000.0, 1.0 / 16000.0, 11000.0)
    out = np.zeros((spec.shape[0], r.I), np.float32)
    for i in range(spec.shape[0]):
        L.lldo_specresample_frame(C.byref(r), spec[i].ctypes.data, out[i].ctypes.data)
    L.lldo_specresample_free(C.byref(r))
    return out


def egemaps_lpc_rows(x, p=11):
    L = lib()
    L.lldo_lpc_acf.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_void_p]
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros((x.shape[0], p), np.float32)
    for i in range(x.shape[0]):
        L.lldo_lpc_acf(x[i].ctypes.data, x.shape[1], p, out[i].ctypes.data)
    return out


def egemaps_formant_rows(lpc):
    L = lib()
    L.lldo_formant_lpc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    lpc = np.ascontiguousarray(lpc, dtype=np.float32)
    out = np.zeros((lpc.shape[0], 10), np.float32)
    roots = np.zeros(2 * lpc.shape[1], np.float64)
    for i in range(lpc.shape[0]):
        L.lldo_formant_lpc(lpc[i].ctypes.data, lpc.shape[1], 5, 1.0 / 11000.0, 50.0, 5450.0, roots.ctypes.data, out[i].ctypes.data)
    return out


def egemaps_harmonics_rows(f0, formants, mag, fs_sec=1024 / 16000.0):
    L = lib()
    L.lldo_harmonics_frame.argtypes = [C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_long, C.c_double, C.c_void_p]
    formants = np.ascontiguousarray(formants, dtype=np.float32)
    mag = np.ascontiguousarray(mag, dtype=np.float32)
    out = np.zeros((mag.shape[0], 6), np.float32)
    for i in range(mag.shape[0]):
        L.lldo_harmonics_frame(float(f0[i]), formants[i].ctypes.data, 5, mag[i].ctypes.data, mag.shape[1], fs_sec, out[i].ctypes.data)
    return out


def frames_cfg(frame_size_sec, win):
    """MfccCfg of a framer + window + FFT front end as the GeMAPS / ComParE sets configure it (no pre-emphasis, symmetric
    zero padding); use with mfcc_chain(cfg, pcm, taps=True) to get the 'fft' / 'mag' levels."""
    c = default_cfg()
    c.frame_size_sec = frame_size_sec
    c.preemph_enable = 0
    c.zero_pad_symmetric = 1
    c.win_func = WIN[win]
    c.win_sigma = 0.4
    c.lofreq = 20.0
    c.n_delta = 0
    return c


# ---------------------------------------------------------------- INTERSPEECH 2010-2012 components (oracle/lld_oracle_is10.c)
IS10_TAPS_CONF = os.path.join(HERE, "conf", "is10_taps.conf")
IS10_LEVELS = ["is10_frames", "is10_intens", "is10_fftc", "is10_outpR", "is10_lpc", "is10_lsp", "is10_pitchShs", "is10_pitch",
               "is10_pitchF", "is10_mspec2", "is10_mspec2log", "is10_jitter", "is10_lld1", "is10_lld2", "is10_lld1_de",
               "is10_lld2_de", "is10_functOnsets", "is10_funct", "is10_functNz"]
VOP = {"add": 0, "mul": 1, "log": 2, "lgA": 3, "sqr": 4, "ee": 5, "abs": 6, "dBp": 7, "dBv": 8, "sum": 9, "ssm": 10, "ll1": 11, "ll2": 12}


def run_reference_is10(pcm, fs=16000, levels=IS10_LEVELS):
    """Real SMILExtract on the unmodified IS10_paraling.conf (+ HTK taps, oracle/conf/is10_taps.conf): {'lld': T x 76
    (lld | lld_de as -lldhtkoutput writes them), 'func': 1 x 1582, level name: its rows}."""
    exe = os.path.join(REF_DIR, "SMILExtract")
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        wav = os.path.join(td, "in.wav")
        write_wav(wav, pcm, fs)
        subprocess.run([exe, "-C", IS10_TAPS_CONF, "-I", wav, "-lldhtkoutput", "lld.htk", "-htkoutput", "func.htk", "-l", "0"],
                       check=True, cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = {}
        for k, fn in (("lld", "lld.htk"), ("func", "func.htk")):
            p = os.path.join(td, fn)
            out[k] = read_htk(p)[0] if os.path.exists(p) else np.zeros((0, 0), np.float32)
        for k in levels:
            p = os.path.join(td, "tap_%s.htk" % k)
            out[k] = read_htk(p)[0] if os.path.exists(p) else np.zeros((0, 0), np.float32)
        return out


def intensity_rows(frames, intensity=0, loudness=1):
    """cIntensity over rows of samples: n x N -> n x (intensity + loudness)."""
    L = lib()
    frames = np.ascontiguousarray(frames, dtype=np.float32)
    n, N = frames.shape
    win = np.zeros(N, np.float64)
    ws = C.c_double(0.0)
    L.lldo_intensity_window.argtypes = [C.c_long, C.c_void_p, C.POINTER(C.c_double)]
    L.lldo_intensity_window(N, win.ctypes.data, C.byref(ws))
    L.lldo_intensity_frame.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_double, C.c_int, C.c_int, C.c_void_p]
    out = np.zeros((n, int(bool(intensity)) + int(bool(loudness))), np.float32)
    for i in range(n):
        L.lldo_intensity_frame(frames[i].ctypes.data, N, win.ctypes.data, N, ws, intensity, loudness, out[i].ctypes.data)
    return out


def lsp_rows(lpc):
    """cLsp over rows of LP coefficients: n x p -> n x p line spectral frequencies."""
    L = lib()
    lpc = np.ascontiguousarray(lpc, dtype=np.float32)
    out = np.zeros_like(lpc)
    L.lldo_lsp_frame.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    for i in range(lpc.shape[0]):
        L.lldo_lsp_frame(lpc[i].ctypes.data, lpc.shape[1], out[i].ctypes.data)
    return out


class _PitchSmoother(C.Structure):
    _fields_ = [("n_cand", C.c_int), ("octave_correction", C.c_int), ("post_simple", C.c_int), ("flags", C.c_int),
                ("voicing_cutoff", C.c_float), ("first_frame", C.c_int), ("ons_flag", C.c_int), ("ons_flag_o", C.c_int),
                ("last_voice", C.c_float), ("last_final", C.c_float), ("pitch_env", C.c_float)]


def pitch_smoother_rows(cands, n_cand=6, voicing_cutoff=0.7, octave_correction=0, post_simple=1, flags=1):
    """cPitchSmoother over the frames of one stream: rows [F0Cand | candVoicing | candScore] (3 n_cand values each) ->
    the rows it writes (flags: 1 F0final, 2 F0finEnv, 4 voicingFinalClipped, 8 voicingFinalUnclipped)."""
    L = lib()
    cands = np.ascontiguousarray(cands, dtype=np.float32)
    s = _PitchSmoother()
    L.lldo_pitch_smoother_init.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
    L.lldo_pitch_smoother_init(C.byref(s), n_cand, voicing_cutoff, octave_correction, post_simple, flags)
    L.lldo_pitch_smoother_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    w = bin(flags & 15).count("1")
    out = []
    row = np.zeros(max(w, 1), np.float32)
    for i in range(cands.shape[0]):
        if L.lldo_pitch_smoother_frame(C.byref(s), cands[i].ctypes.data, row.ctypes.data) > 0:
            out.append(row[:w].copy())
    return np.array(out, np.float32).reshape(-1, w)


def vecop_rows(x, op, param1=1.0, logfloor=1e-12):
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros_like(x)
    L.lldo_vecop.argtypes = [C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_long]
    assert L.lldo_vecop(VOP[op], param1, logfloor, x.ctypes.data, out.ctypes.data, x.size) == 0
    return out


def specresample_rows(spec, fs_sec, last_fs_sec, base_period, target_fs):
    """cSpecResample for any geometry: n x Nfft packed spectra -> n x n_out samples."""
    L = lib()
    L.lldo_specresample_init.argtypes = [C.POINTER(_SpecRes), C.c_long, C.c_double, C.c_double, C.c_double, C.c_double]
    L.lldo_specresample_frame.argtypes = [C.POINTER(_SpecRes), C.c_void_p, C.c_void_p]
    L.lldo_specresample_free.argtypes = [C.POINTER(_SpecRes)]
    spec = np.ascontiguousarray(spec, dtype=np.float32)
    r = _SpecRes()
    L.lldo_specresample_init(C.byref(r), spec.shape[1], fs_sec, last_fs_sec, base_period, target_fs)
    out = np.zeros((spec.shape[0], r.I), np.float32)
    for i in range(spec.shape[0]):
        L.lldo_specresample_frame(C.byref(r), spec[i].ctypes.data, out[i].ctypes.data)
    L.lldo_specresample_free(C.byref(r))
    return out


def vecop_reduce_rows(x, op):
    """cVectorOperation's vector-to-scalar operations (sum, ssm, ll1, ll2) over rows: n x N -> n."""
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.float32)
    L.lldo_vecop_reduce.restype = C.c_float
    L.lldo_vecop_reduce.argtypes = [C.c_int, C.c_void_p, C.c_long]
    return np.array([L.lldo_vecop_reduce(VOP[op], x[i].ctypes.data, x.shape[1]) for i in range(x.shape[0])], np.float32)


def mfcc_inverse_rows(mfcc, first, last, n_bands, cep_lifter, htk=1, do_log=1):
    """cMfcc with inverse = 1 (oracle/lld_oracle_compare.c::lldo_mfcc_inverse): n x (last - first + 1) cepstra -> n x n_bands"""
    L = lib()
    x = np.ascontiguousarray(mfcc, dtype=np.float32)
    assert x.shape[1] == last - first + 1
    L.lldo_mfcc_inverse.restype = None
    L.lldo_mfcc_inverse.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]
    out = np.zeros((x.shape[0], n_bands), np.float32)
    for i in range(x.shape[0]):
        L.lldo_mfcc_inverse(x[i].ctypes.data, first, last, n_bands, cep_lifter, htk, do_log, out[i].ctypes.data)
    return out


def melspec_inverse_rows(mel, n_out, frame_size_sec, lofreq=0.0, hifreq=8000.0, use_power=1, htk=1, tables=False):
    """cMelspec with inverse = 1 (oracle/lld_oracle_compare.c::lldo_melspec_inverse): n x n_src mel bands -> n x n_out spectrum bins;
    tables = True: also (nLoF, nHiF, weights[n_out], channels[n_out] as int32), the tables of the bank"""
    L = lib()
    x = np.ascontiguousarray(mel, dtype=np.float32)
    L.lldo_melspec_inverse.restype = None
    L.lldo_melspec_inverse.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    out = np.zeros((x.shape[0], n_out), np.float32)
    tab = np.zeros(2 + 2 * n_out, np.float32)
    for i in range(x.shape[0]):
        L.lldo_melspec_inverse(x[i].ctypes.data, x.shape[1], n_out, frame_size_sec, lofreq, hifreq, use_power, htk, out[i].ctypes.data,
                               tab.ctypes.data if i == 0 else None)
    if not tables:
        return out
    if x.shape[0] == 0:
        L.lldo_melspec_inverse(np.zeros(x.shape[1], np.float32).ctypes.data, x.shape[1], n_out, frame_size_sec, lofreq, hifreq, use_power, htk,
                               np.zeros(n_out, np.float32).ctypes.data, tab.ctypes.data)
    return out, (int(tab[0]), int(tab[1]), tab[2:2 + n_out].copy(), tab[2 + n_out:].astype(np.int32))


def plp_static_stage(pcm, stage):
    """[plp:cPlp]'s level of config/plp/PLP_0_D_A.conf cut after a stage: 1 = autocorrelation (doLP = 0), 2 = LP coefficients
    (doLpToCeps = 0), 3 = cepstra (as shipped)"""
    L = lib()
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    L.lldo_plp_static_stage.restype = C.c_long
    L.lldo_plp_static_stage.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_void_p]
    T = L.lldo_plp_static_stage(pcm.ctypes.data, len(pcm), stage, None)
    out = np.zeros((max(T, 0), 5 if stage == 2 else 6), np.float32)
    if T > 0:
        L.lldo_plp_static_stage(pcm.ctypes.data, len(pcm), stage, out.ctypes.data)
    return out


def plp_stage_rows(mel, band_hz, lp_order, compression, stage, cep_lifter=22):
    """cPlp in HTK mode up to a stage (oracle/lld_oracle_compare.c::lldo_plp_stage): 1 = autocorrelation (doLP = 0), 2 = LP coefficients
    (doLpToCeps = 0), 3 = cepstra. mel: n x n_bands; band_hz: the bands' centre frequencies (the level's meta data)."""
    L = lib()
    mel = np.ascontiguousarray(mel, dtype=np.float32)
    hz = np.ascontiguousarray(band_hz, dtype=np.float64)
    n_out = {1: lp_order + 1, 2: lp_order, 3: lp_order + 1}[stage]
    L.lldo_plp_stage.restype = None
    L.lldo_plp_stage.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]
    out = np.zeros((mel.shape[0], n_out), np.float32)
    for i in range(mel.shape[0]):
        L.lldo_plp_stage(mel[i].ctypes.data, mel.shape[1], hz.ctypes.data, lp_order, compression, cep_lifter, stage, out[i].ctypes.data)
    return out


class _Spectral(C.Structure):
    _fields_ = [("K", C.c_long), ("fsSec", C.c_double), ("prev", C.c_void_p), ("have_prev", C.c_int), ("frq", C.c_void_p),
                ("sharp", C.c_void_p)]


class SpectralOpts(C.Structure):
    """lldo_spectral_opts (oracle/lld_oracle.h)"""
    _fields_ = [("n_bands", C.c_int), ("band_lo", C.c_long * 16), ("band_hi", C.c_long * 16), ("n_rolloff", C.c_int),
                ("rolloff", C.c_double * 16)] + [(k, C.c_int) for k in ("flux", "centroid", "max_pos", "min_pos", "entropy", "variance",
                                                                        "skewness", "kurtosis", "slope", "sharpness", "harmonicity", "flatness",
                                                                        "log_flatness", "spec_diff", "spec_pos_diff", "flux_centroid",
                                                                        "flux_at_flux_centroid", "standard_deviation", "n_slopes")] + [
                   ("slope_lo", C.c_long * 16), ("slope_hi", C.c_long * 16)]


def spectral_general_rows(mag, frame_size_sec, bands, rolloff=(0.25, 0.5, 0.75, 0.9), slopes=(), **flags):
    """cSpectral over the frames of one stream for any of the shipped descriptor sets: n x K magnitudes -> n x count."""
    L = lib()
    mag = np.ascontiguousarray(mag, dtype=np.float32)
    o = SpectralOpts()
    o.n_bands = len(bands)
    for i, (a, b) in enumerate(bands):
        o.band_lo[i], o.band_hi[i] = a, b
    o.n_rolloff = len(rolloff)
    for i, r in enumerate(rolloff):
        o.rolloff[i] = r
    o.n_slopes = len(slopes)
    for i, (a, b) in enumerate(slopes):
        o.slope_lo[i], o.slope_hi[i] = a, b
    for k, v in flags.items():
        setattr(o, k, int(v))
    s = _Spectral()
    L.lldo_spectral_init.argtypes = [C.c_void_p, C.c_long, C.c_double]
    L.lldo_spectral_general.restype = C.c_int
    L.lldo_spectral_general.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.lldo_spectral_free.argtypes = [C.c_void_p]
    L.lldo_spectral_init(C.byref(s), mag.shape[1], frame_size_sec)
    row = np.zeros(64, np.float32)
    out = []
    for i in range(mag.shape[0]):
        n = L.lldo_spectral_general(C.byref(s), C.byref(o), mag[i].ctypes.data, row.ctypes.data)
        out.append(row[:n].copy())
    L.lldo_spectral_free(C.byref(s))
    return np.array(out, np.float32)


class _SpecScale(C.Structure):
    _fields_ = [("K", C.c_long), ("ft", C.c_void_p), ("sigma", C.c_void_p), ("d1", C.c_void_p), ("d2", C.c_void_p), ("k", C.c_void_p),
                ("co", C.c_void_p), ("audw", C.c_void_p), ("meta", C.c_float * 8)]


class _Shs(C.Structure):
    _fields_ = [("N", C.c_long), ("n_octaves", C.c_float), ("points_per_octave", C.c_float), ("Fmint", C.c_float), ("Fstept", C.c_float),
                ("base", C.c_double), ("n_harmonics", C.c_int), ("compression", C.c_float), ("n_cand", C.c_int), ("min_pitch", C.c_double),
                ("max_pitch", C.c_double), ("voicing_cutoff", C.c_float), ("old_peaks", C.c_int)]


def specscale_shs_rows(mag, frame_size_sec, min_f=25.0, flags=7, n_cand=6, old_peaks=0, voicing_cutoff=0.7, n_harmonics=15,
                       compression=0.85, min_pitch=52.0, max_pitch=620.0):
    """cSpecScale (flags: 1 specEnhance, 2 specSmooth, 4 auditoryWeighting) then cPitchShs over rows of K magnitudes:
    (n x K octave-scale spectra, n x (3 n_cand + 3) rows [nCandidates | F0Cand | candVoicing | candScores | F0raw | voicingClip])."""
    L = lib()
    mag = np.ascontiguousarray(mag, dtype=np.float32)
    n, K = mag.shape
    s, h = _SpecScale(), _Shs()
    L.lldo_specscale_init_ex.restype = C.c_int
    L.lldo_specscale_init_ex.argtypes = [C.c_void_p, C.c_long, C.c_double, C.c_double]
    L.lldo_specscale_frame_ex.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.lldo_shs_init.argtypes = [C.c_void_p, C.c_void_p]
    L.lldo_pitch_shs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.lldo_specscale_free.argtypes = [C.c_void_p]
    assert L.lldo_specscale_init_ex(C.byref(s), K, frame_size_sec, min_f) == 1
    L.lldo_shs_init(C.byref(h), C.byref(s))
    h.n_cand, h.old_peaks, h.voicing_cutoff, h.n_harmonics, h.compression = n_cand, old_peaks, voicing_cutoff, n_harmonics, compression
    h.min_pitch, h.max_pitch = min_pitch, max_pitch
    hps = np.zeros((n, K), np.float32)
    shs = np.zeros((n, 3 * n_cand + 3), np.float32)
    for i in range(n):
        L.lldo_specscale_frame_ex(C.byref(s), flags, mag[i].ctypes.data, hps[i].ctypes.data)
        L.lldo_pitch_shs(C.byref(h), hps[i].ctypes.data, shs[i].ctypes.data, None)
    L.lldo_specscale_free(C.byref(s))
    return hps, shs


# ---------------------------------------------------------------- IS10_paraling as a whole chain (the oracle of a future fused chain)
def window_chain(x, kinds, Ws):
    """lldo_window_chain: tick-accurate chain of window processors over ONE level (kinds: 0 delta, 1 SMA, 2 SMA with noZeroSma,
    3 delta with onlyInSegments); returns the levels, stage s with rows + sum(W[:s+1]) rows."""
    L = lib()
    x = np.ascontiguousarray(x, np.float32)
    T, D = x.shape
    outs, tot = [], T
    for w in Ws:
        tot += w
        outs.append(np.zeros((tot, D), np.float32))
    if T <= 0:
        return [o[:0] for o in outs]
    ptrs = (C.c_void_p * len(Ws))(*[a.ctypes.data for a in outs])
    L.lldo_window_chain.restype = None
    L.lldo_window_chain.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.lldo_window_chain(x.ctypes.data, T, D, len(Ws), (C.c_int * len(Ws))(*kinds), (C.c_int * len(Ws))(*Ws), ptrs)
    return outs


def _is10_cfg25(n_bands, hifreq, last):
    c = default_cfg()
    c.frame_size_sec, c.preemph_enable, c.preemph_k, c.win_func, c.zero_pad_symmetric = 0.025, 1, 0.97, WIN["ham"], 1
    c.n_bands, c.lofreq, c.hifreq, c.use_power, c.mel_htk_compatible = n_bands, 20.0, hifreq, 1, 0
    c.first_mfcc, c.last_mfcc, c.cep_lifter, c.mfcc_htk_compatible, c.n_delta = 0, last, 22.0, 0, 0
    return c


def is10_levels(pcm):
    """config/is09-13/IS10_paraling.conf from the samples on (16 kHz): every level up to lld1 / lld2 and their deltas, as the binary
    holds them (pinned by tests/test_oracle_pin_is10.py). Needs at least two 60 ms frames."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    mf, t = mfcc_chain(_is10_cfg25(26, 8000.0, 14), pcm, taps=True)
    _, t8 = mfcc_chain(_is10_cfg25(8, 6500.0, 7), pcm, taps=True)
    ml = vecop_rows(t8["mel"], "log")
    lsp = lsp_rows(egemaps_lpc_rows(specresample_rows(t["fft"], 512 / 16000.0, 400 / 16000.0, 1 / 16000.0, 11000.0), 8))
    x = (pcm.astype(np.float32) / np.float32(32767.0)).astype(np.float32)
    T25 = mf.shape[0]
    inten = intensity_rows(np.stack([x[i * 160:i * 160 + 400] for i in range(T25)]), 0, 1)
    c60 = frames_cfg(0.060, "gauss")
    c60.win_sigma = 0.25
    _, t60 = mfcc_chain(c60, pcm, taps=True)
    _, shs = specscale_shs_rows(t60["mag"], 1024 / 16000.0, 20.0, 7, 6, 0, 0.7)
    pitch = pitch_smoother_rows(shs[:, 1:19], flags=2 | 8)
    pitchF = pitch_smoother_rows(shs[:, 1:19], flags=1)
    L = lib()
    L.lldo_pitch_jitter_ex.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_double, C.c_double, C.c_double,
                                       C.c_void_p, C.c_void_p]
    L.lldo_set_jitter_time_shift.argtypes = [C.c_long]
    f0 = np.ascontiguousarray(pitchF.reshape(-1))
    jit = np.zeros((len(f0), 4), np.float32)
    if len(f0):
        compare_set_is13(True)                               # useBrokenJitterThresh: the option's default, 1
        L.lldo_set_jitter_time_shift(1)
        try:
            L.lldo_pitch_jitter_ex(x.ctypes.data, len(x), f0.ctypes.data, len(f0), 960, 160, 16000.0, 0.010, 0.2, jit.ctypes.data, None)
        finally:
            compare_set_is13(False)
            L.lldo_set_jitter_time_shift(0)
    Tm = pitch.shape[0]                                      # the shortest input level decides: T60 - 1 rows
    # [is10_lld]: SMA(3) over five levels of different lengths -- the 25 ms levels hold real frames beyond Tm, the pitch level is padded
    s25 = window_chain(np.concatenate([inten, mf, ml, lsp], axis=1), [1], [1])[0]
    sp = window_chain(pitch, [1], [1])[0]
    lld1 = np.concatenate([s25[:Tm + 1], sp], axis=1)
    lld1_de = window_chain(lld1, [0], [2])[0]
    # [is10_lld2] (noZeroSma) and [is10_delta2] (onlyInSegments: its norm grows over the whole file)
    lld2, lld2_de = window_chain(np.concatenate([pitchF, jit[:, :3]], axis=1), [2, 3], [1, 2])
    return {"lld1": lld1, "lld1_de": lld1_de, "lld2": lld2, "lld2_de": lld2_de, "pitchF": pitchF}


def is10_lld_chain(pcm):
    """The 76 columns -lldhtkoutput writes for IS10_paraling.conf: [lld1 | lld2 | lld1_de | lld2_de], T60 rows."""
    d = is10_levels(pcm)
    n = d["lld1"].shape[0]
    return np.concatenate([d["lld1"], d["lld2"][:n], d["lld1_de"][:n], d["lld2_de"][:n]], axis=1)


def _is10_spec_l1(nz):
    """[is10_functL1] / [is10_functL1nz] of IS10_paraling_core.func.conf.inc"""
    s = FuncSpec()
    _spec_common(s, ["Extremes", "Regression", "Moments", "Percentiles", "Times"])
    s.ext_mask, s.ext_norm = _mask(EXT_NAMES, ["maxPos", "minPos", "amean"]), NORM["frame"]
    s.reg_mask = _mask(REG_NAMES, ["linregc1", "linregc2", "linregerrA", "linregerrQ"])
    s.mom_mask = _mask(MOM_NAMES, ["stddev", "skewness", "kurtosis"])
    s.pct_mask, s.pct_interp = 0x3f, 1
    if nz:
        s.n_pctl, s.n_range, s.non_zero_functs = 1, 0, 1
        s.pctl[0] = 0.99
    else:
        s.n_pctl, s.n_range = 2, 1
        s.pctl[0], s.pctl[1] = 0.01, 0.99
        s.range_a[0], s.range_b[0] = 0, 1
    s.times_mask, s.times_norm = _mask(TIMES_NAMES, ["upleveltime75", "upleveltime90"]), NORM["segment"]
    return s


def is10_func(pcm):
    """The 1582 functionals of IS10_paraling.conf: [is10_functL1 (34 x 2 x 21) | is10_functL1nz (4 x 2 x 19) | is10_functOnsets (2)].
    The first two instances summarise the first T - 3 rows of their (smoothed ; delta) levels (T = rows of the smoothed level; measured
    against the binary), the onsets instance every row of is10_pitchF."""
    d = is10_levels(pcm)
    T = d["lld1"].shape[0]
    n = max(T - 3, 1)
    f1 = funcspec(np.concatenate([d["lld1"][:n], d["lld1_de"][:n]], axis=1), _is10_spec_l1(False)).reshape(1, -1)
    f2 = funcspec(np.concatenate([d["lld2"][:n], d["lld2_de"][:n]], axis=1), _is10_spec_l1(True)).reshape(1, -1)
    so = FuncSpec()
    _spec_common(so, ["Onset", "Times"])
    so.ons_mask, so.ons_norm, so.times_mask, so.times_norm = 1 << 4, NORM["segment"], 1 << 12, NORM["second"]
    f3 = funcspec(d["pitchF"], so).reshape(1, -1)
    return np.concatenate([f1, f2, f3], axis=1)
