"""Pin the functionals part of the CPU oracle (oracle/lld_oracle_func.c) against the func
level of the REAL reference binary (config/is09-13/IS09_emotion.conf, -htkoutput): 32 LLD
columns x 12 functionals = 384 values per utterance, bit-exact; and the rule for how many
LLD rows the full-mode functionals summarise (max(1, T-2) of the T+1 rows in the LLD file)."""
import numpy as np
import pytest

KEYS = ["u2_32000", "u3_16000", "u10_16000", "u1_16000", "u0_16000", "u7_400", "u7_560", "u5_160000"]


def func_rows(lld_rows):
    return max(1, lld_rows - 3) if lld_rows > 0 else 0


@pytest.mark.parametrize("key", KEYS)
def test_is09_functionals_bit_exact(oracle, golden_func, key):
    lld, ref = golden_func["lld_" + key], golden_func["func_" + key]
    assert ref.shape == (1, 384) and lld.shape[1] == 32
    out = oracle.functionals(lld[:func_rows(lld.shape[0])]).reshape(1, -1)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), f"max abs {np.abs(out - ref).max()}"


def test_functionals_known_answers(oracle):
    """Hand-checkable values: a ramp and a constant."""
    x = np.stack([np.arange(10, dtype=np.float32), np.full(10, 2.5, np.float32)], axis=1)
    f = oracle.functionals(x)
    names = ["max", "min", "range", "maxpos", "minpos", "amean", "linregc1", "linregc2", "linregerrQ", "stddev",
             "skewness", "kurtosis"]
    ramp = dict(zip(names, f[0]))
    assert (ramp["max"], ramp["min"], ramp["range"], ramp["maxpos"], ramp["minpos"]) == (9, 0, 9, 9, 0)
    assert ramp["amean"] == 4.5 and abs(ramp["linregc1"] - 1) < 1e-6 and abs(ramp["linregc2"]) < 1e-6
    assert ramp["linregerrQ"] < 1e-10 and abs(ramp["stddev"] - np.sqrt(8.25)) < 1e-6 and abs(ramp["skewness"]) < 1e-6
    const = dict(zip(names, f[1]))
    assert const["stddev"] == 0 and const["skewness"] == 0 and const["kurtosis"] == 0 and abs(const["linregc1"]) < 1e-12
    assert const["linregc2"] == 2.5 and const["maxpos"] == 0 and const["minpos"] == 0


def test_functionals_single_row(oracle):
    x = np.array([[3.0, -1.0]], np.float32)
    f = oracle.functionals(x)
    assert f[0, 0] == 3.0 and f[0, 7] == 3.0 and f[1, 7] == -1.0      # max, linregc2 = the value itself
    assert f[0, 6] == 0.0 and f[0, 9] == 0.0
