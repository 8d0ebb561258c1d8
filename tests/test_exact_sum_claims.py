"""The two "exact in any order" arguments the kernels rely on, checked as properties on the host.

lld_f0_jitter: the means of a cross-correlation are double sums of float samples s / 32767 (|x| <= 32768 / 32767 < 2, multiples
of 2^-38): any partial sum of up to 4096 of them is below 2^13 and a multiple of 2^-38 -- 51 bits -- hence exact, so a reduction tree + scan replaces the reference's sequential pass bit for bit.
lld_f0_cand: the mean of the summation spectrum (513 non-negative floats, widened) is formed as a tree sum when
exponent(S) - exponent(smallest non-zero value) <= 28; then every partial sum of every order is representable."""
import math

import numpy as np
from hypothesis import given, settings, strategies as st


def seq_sum(x):
    s = 0.0
    for v in x:
        s += float(v)
    return s


def tree_sum(x):
    x = [float(v) for v in x]
    while len(x) > 1:
        x = [x[i] + x[i + 1] if i + 1 < len(x) else x[i] for i in range(0, len(x), 2)]
    return x[0] if x else 0.0


def pcm_to_float(s):
    # lld_device.hpp pcm16_to_float == the correctly rounded float of s / 32767 (test_host_logic pins that)
    return np.float32(np.float64(s) / 32767.0)


@settings(max_examples=200, deadline=None)
@given(st.lists(st.integers(-32768, 32767), min_size=1, max_size=640), st.randoms(use_true_random=False))
def test_sums_of_pcm_floats_are_exact_in_any_order(samples, rnd):
    x = [pcm_to_float(s) for s in samples]
    exact = math.fsum(float(v) for v in x)              # the real-number sum, correctly rounded ...
    assert seq_sum(x) == exact                          # ... is what the sequential double sum gives: no addition rounded
    y = list(x)
    rnd.shuffle(y)
    assert seq_sum(y) == exact and tree_sum(y) == exact and tree_sum(x) == exact
    # every sample is a multiple of 2^-38 of magnitude < 2 (what the argument in the kernel's comment uses)
    for v in x:
        assert abs(float(v)) < 2.0 and float(v) * 2.0 ** 38 == math.floor(float(v) * 2.0 ** 38)


def cand_condition(vals):
    """the test lld_f0_cand applies (f0_shs, mean_exact)"""
    nz = [float(v) for v in vals if v > 0]
    s = tree_sum(vals)
    if s == 0.0:
        return True
    mn = min(nz)
    if mn < 1.17549435e-38:
        return False
    return math.frexp(s)[1] - math.frexp(mn)[1] <= 28


def test_tree_mean_equals_the_sequential_chain_when_the_condition_holds():
    rng = np.random.default_rng(17)
    held = 0
    for case in range(400):
        spread = int(rng.integers(0, 41))               # dynamic range 2^spread: both outcomes of the condition occur
        scale = float(10.0 ** rng.uniform(-20, 20))
        vals = (scale * 2.0 ** (-spread * rng.random(513))).astype(np.float32)
        vals[rng.random(513) < 0.1] = 0.0
        if not cand_condition(vals):
            continue                                    # the kernel runs the sequential chain: nothing to prove
        held += 1
        exact = math.fsum(float(v) for v in vals)
        assert seq_sum(vals) == exact and tree_sum(vals) == exact, (case, spread)
        y = rng.permutation(vals)
        assert seq_sum(y) == exact and tree_sum(y) == exact
    assert 100 < held < 400


def test_the_condition_is_tight_enough_to_matter():
    # one ulp-sized straggler beyond the bound: the sequential sum rounds, and the condition says so
    vals = [np.float32(1.0)] * 512 + [np.float32(2.0 ** -30)]
    assert not cand_condition(vals)
    vals = [np.float32(1.0)] * 512 + [np.float32(2.0 ** -18)]
    assert cand_condition(vals) and seq_sum(vals) == math.fsum(float(v) for v in vals)


def test_float_division_by_a_known_divisor_in_five_operations(tmp_path):
    """oo_quad_irfft_even_real divides the inverse transform's outputs by the frame's one divisor (cAcf: the number of bins) with
    y = RN(1 / b) and two residual corrections instead of the division instruction sequence. Every dividend significand, three
    binades, both signs, for the divisors the shipped geometries produce (129 .. 2049) and some hostile ones."""
    import os
    import subprocess
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "markstein_f32.c")
    exe = str(tmp_path / "markstein_f32")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, src, "-lm"], check=True)
    out = subprocess.run([exe, "257", "513", "129", "1025", "2049", "3", "1.9999999", "1.0000001", "16777215"], check=True,
                         capture_output=True, text=True).stdout
    assert int(out) == 0


def test_float_division_by_the_number_of_harmonics_in_five_operations(tmp_path):
    """lld_f0_cand's f0_shs (round 6) divides the summation spectrum by nHarmonics the same way -- y = RN(1 / b), a product, two residual
    corrections -- when no value of the frame lies in (0, 2^-100). Every dividend significand for every divisor 1 .. 32 (the same
    statement on the device's own instructions, with ten exponents of the dividend: tools/ubench/div_f32_by_const_check.hip,
    profiles/r06_div_f32_by_const_check.json)."""
    import os
    import subprocess
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "markstein_f32.c")
    exe = str(tmp_path / "markstein_f32")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, src, "-lm"], check=True)
    out = subprocess.run([exe] + [str(b) for b in range(1, 33)], check=True, capture_output=True, text=True).stdout
    assert int(out) == 0

