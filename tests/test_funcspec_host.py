"""Host-side logic of the general functionals API (no GPU needed): spec layout shared by the C ABI and the oracle, value
counts of the ComParE_2016 / IS13_ComParE presets, validation messages, and that compute entry points refuse to run
without a device instead of falling back."""
import ctypes as C

import numpy as np
import pytest

from opensmile_amd import capi

EXPECTED = {"A": 31, "B": 31, "F0": 5, "Nz": 39, "LLD": 23, "Delta": 15}


def test_spec_layout_matches_oracle(oracle):
    assert C.sizeof(capi.FuncSpec) == C.sizeof(oracle.FuncSpec) == 784
    assert [f for f, _ in capi.FuncSpec._fields_] == [f for f, _ in oracle.FuncSpec._fields_]


@pytest.mark.parametrize("inst", sorted(EXPECTED))
def test_preset_counts_and_oracle_agreement(oracle, inst):
    for mine, theirs in ((capi.funcspec_compare16(inst), oracle.compare16_func_spec(inst)),
                         (capi.funcspec_is13_compare(inst), oracle.is13_func_spec(inst))):
        assert capi.funcspec_count(mine) == EXPECTED[inst] == len(oracle.funcspec_names(theirs))
        # every field a family of the instance reads is identical (unused families' norms may differ)
        o = oracle.FuncSpec()
        C.memmove(C.byref(o), C.byref(mine), C.sizeof(mine))
        assert oracle.funcspec_names(o) == oracle.funcspec_names(theirs)
        x = np.random.default_rng(1).standard_normal((50, 3)).astype(np.float32)
        assert np.array_equal(oracle.funcspec(x, o), oracle.funcspec(x, theirs))
    total = sum(EXPECTED[i] * n for i, n in (("A", 8), ("B", 110), ("Nz", 12), ("F0", 1), ("LLD", 59), ("Delta", 59)))
    assert total == 6373 == capi.load().smilehip_functionals_compare16_count()


def test_validation_messages():
    s = capi.funcspec_compare16("Nz")
    s.n_pctl = 9
    with pytest.raises(capi.SmileHipError, match="at most 8"):
        capi.funcspec_count(s)
    s = capi.funcspec_compare16("LLD")
    s.reg_norm_coeff = 3
    with pytest.raises(capi.SmileHipError, match="normRegCoeff"):
        capi.funcspec_count(s)
    s = capi.funcspec_compare16("A")
    s.times_norm = 7
    with pytest.raises(capi.SmileHipError, match="time norm"):
        capi.funcspec_count(s)
    s = capi.funcspec_compare16("A")
    s.n_fam = 0
    with pytest.raises(capi.SmileHipError, match="functionalsEnabled"):
        capi.funcspec_count(s)
    s = capi.funcspec_compare16("A")
    s.fam[0] = 42
    with pytest.raises(capi.SmileHipError, match="unknown functional family"):
        capi.funcspec_count(s)


def test_no_cpu_fallback_without_device():
    """A host-only plan (tables, geometry) cannot hold a batch, let alone run functionals: SMILEHIP_ERR_NO_DEVICE, never
    a CPU path."""
    plan = capi.Plan(None, capi.compare16_config())
    with pytest.raises(capi.SmileHipError, match="no device"):
        capi.Batch(plan, np.array([0, 16000], np.int64))
    L = capi.load()
    s = capi.funcspec_compare16("A")
    assert L.smilehip_funcspec_matrix(None, C.byref(s), None, 4, 10, 4, None, None) != 0
