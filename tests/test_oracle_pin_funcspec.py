"""Pins oracle/lld_oracle_funcspec.c (all nine functional families, ComParE_2016's six cFunctionals instances) against
the real binary: golden 6373-value functionals vectors of SMILExtract on ComParE_2016.conf, computed by the oracle from
exactly the level rows the reference's functionals saw (HTK taps, oracle/conf/compare_func_taps.conf) -- bit for bit.

Which rows a full-mode cFunctionals summarises is decided by its first end-of-input tick (winToVecProcessor.cpp:504-528,
868-1098), measured here against the binary (T = frames of the 60 ms framer, P = frames the Viterbi smoother had not
decided at end of input):
    A     (lldA_smo;lldA_smo_de)   T-2          B   (lldB_smo;lldB_smo_de)   T+2 (one row more than the LLD sinks keep)
    LLD   (lldA_smo;lldB_smo)      T            Delta (…_de levels)          T-2
    F0    (lld_f0_nzsmo)           T-P          Nz  (lld_nzsmo;lld_nzsmo_de) T-P-2         [P >= T: T and T-2]"""
import ctypes as C
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KEYS = ["u3_48000", "u5_16000", "u4_9000", "u37_9000", "u2_8720", "u7_2720", "u7_1760", "u7_1440"]
ORDER = ["A", "B", "Nz", "F0", "LLD", "Delta"]          # writer levels as [functionals] concatenates them


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "compare16_func_synth.npz"))


def pending(oracle, pcm):
    """(T60, P) from the oracle's F0 chain."""
    L = oracle.lib()
    L.lldo_pitch_viterbi.restype = None
    L.lldo_pitch_viterbi.argtypes = [C.c_void_p, C.c_long, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    out, t = oracle.compare_f0_chain(pcm, taps=True)
    T = out.shape[0]
    shs = np.ascontiguousarray(t["shs"])
    P = C.c_long(0)
    tmp = np.zeros((T, 2), np.float32)
    L.lldo_pitch_viterbi(shs.ctypes.data, T, np.float32(0.7), tmp.ctypes.data, None, C.byref(P))
    return T, P.value


def func_rows(inst, T, P):
    p = 0 if P >= T else P
    n = {"A": T - 2, "B": T + 2, "LLD": T, "Delta": T - 2, "F0": T - p, "Nz": T - p - 2}[inst]
    return max(1, n)


def inputs(g, key):
    def cat(*names):
        m = [g[n + "_" + key] for n in names]
        r = min(x.shape[0] for x in m)
        return np.concatenate([x[:r] for x in m], axis=1)
    return {"A": cat("a_smo", "a_de"), "B": cat("b_smo", "b_de"), "F0": g["f0_smo_" + key], "Nz": cat("nz_smo", "nz_de"),
            "LLD": cat("a_smo", "b_smo"), "Delta": cat("a_de", "b_de")}


@pytest.mark.parametrize("key", KEYS)
def test_oracle_functionals_bit_exact_vs_binary(oracle, golden, key):
    T, P = pending(oracle, golden["pcm_" + key])
    X = inputs(golden, key)
    f = golden["func_" + key]
    assert f.shape == (6373,)
    pos = 0
    for inst in ORDER:
        spec = oracle.compare16_func_spec(inst)
        per = len(oracle.funcspec_names(spec))
        cols = X[inst].shape[1]
        ref = f[pos:pos + per * cols].reshape(cols, per)
        pos += per * cols
        n = func_rows(inst, T, P)
        out = oracle.funcspec(np.ascontiguousarray(X[inst][:n]), spec)
        same = out.view(np.uint32) == ref.view(np.uint32)
        names = oracle.funcspec_names(spec)
        assert same.all(), (f"{key}/{inst} (T={T}, P={P}, rows={n}): "
                            f"{[(names[k], int(c)) for c, k in np.argwhere(~same)[:5]]}")
    assert pos == 6373


def test_value_names_match_binary(oracle, golden):
    """The value-name suffixes a spec generates are the ones cFunctionals::setupNamesForElement produced."""
    names = [str(n) for n in golden["names"]]
    assert len(names) == 6373
    spec = oracle.compare16_func_spec("A")
    suffixes = oracle.funcspec_names(spec)
    assert names[:len(suffixes)] == ["audspec_lengthL1norm_sma_" + s for s in suffixes]
    spec = oracle.compare16_func_spec("Delta")
    suffixes = oracle.funcspec_names(spec)
    assert names[-len(suffixes):] == ["mfcc_sma_de[14]_" + s for s in suffixes]
    assert names[4126] == "F0final_sma_ff0_nnz"       # [is13_functionalsF0] has functNameAppend = ff0


def test_funcspec_count_and_rejects(oracle):
    for inst, per in (("A", 31), ("B", 31), ("F0", 5), ("Nz", 39), ("LLD", 23), ("Delta", 15)):
        assert len(oracle.funcspec_names(oracle.compare16_func_spec(inst))) == per
    s = oracle.compare16_func_spec("A")
    s.n_fam = 13
    with pytest.raises(ValueError):
        oracle.funcspec(np.zeros((4, 2), np.float32), s)
