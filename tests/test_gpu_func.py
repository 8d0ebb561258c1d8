"""IS09_emotion end to end on the GPU: LLD chain (smilehip_lld_run) -> functionals
(smilehip_batch_functionals) = the 384-value vector of config/is09-13/IS09_emotion.conf,
against the func level of the real reference binary and against the CPU oracle."""
import numpy as np
import pytest

from test_oracle_pin_func import KEYS

pytestmark = pytest.mark.gpu

NAMES = ["max", "min", "range", "maxpos", "minpos", "amean", "linregc1", "linregc2", "linregerrQ", "stddev", "skewness",
         "kurtosis"]


@pytest.fixture(scope="module")
def hip():
    from opensmile_amd import capi
    ctx = capi.Context(0)
    plan = capi.Plan(ctx, capi.is09_lld_config())
    return capi, ctx, plan


def test_functionals_kernel_vs_oracle_on_reference_lld(hip, oracle, golden_func):
    """Same input (the REAL binary's own LLD matrix) -> the binary's functionals vector, bit for bit: every contour is
    walked in the reference's order with its accumulator types (none of IS09's 12 functionals passes through libm)."""
    capi, ctx, plan = hip
    llds = [golden_func["lld_" + k] for k in KEYS]
    # a batch whose row counts equal the golden LLD row counts: T = rows - 1 frames
    lens = [400 + 160 * (x.shape[0] - 2) for x in llds]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    b = capi.Batch(plan, off)
    np.testing.assert_array_equal(np.diff(b.frame_offsets), [x.shape[0] for x in llds])
    np.testing.assert_array_equal(b.func_rows(), [max(1, x.shape[0] - 3) for x in llds])
    out = b.functionals_host(np.concatenate(llds))
    assert out.shape == (len(KEYS), 384)
    for i, k in enumerate(KEYS):
        ref = golden_func["func_" + k][0]
        o = out[i]
        same = o.view(np.uint32) == ref.view(np.uint32)
        assert same.all(), (k, [NAMES[j % 12] for j in np.flatnonzero(~same)[:6]])
    b.close()


def test_is09_emotion_end_to_end(hip, oracle, golden_func):
    """PCM -> LLD (GPU) -> functionals (GPU) vs the binary's 384 values. The LLD level carries
    the FFT/log round-off of the GPU path (and possible F0 flips), so values are compared on
    each functional's natural scale."""
    capi, ctx, plan = hip
    keys = [k for k in KEYS]
    pcms = [golden_func["pcm_" + k] for k in keys]
    off = np.concatenate([[0], np.cumsum([len(p) for p in pcms])]).astype(np.int64)
    b = capi.Batch(plan, off)
    lld = b.run_host(np.concatenate(pcms))
    f = b.functionals_host(lld)
    for i, k in enumerate(keys):
        ref = golden_func["func_" + k][0].reshape(32, 12)
        o = f[i].reshape(32, 12)
        ref_lld = golden_func["lld_" + k]
        # natural scales of the LLD test (tests/test_gpu_is09.py): MFCC block on the largest cepstral
        # magnitude, deltas on their static column's scale, voicing probability absolute 1e-4
        scale = np.maximum(np.abs(ref_lld[:, :16]).max(axis=0), 1e-12)
        scale[1:13] = max(float(np.abs(ref_lld[:, 1:13]).max()), 1e-12)
        scale[0] = max(scale[0], 1e-3)
        scale[14] = 5.0
        scale = np.concatenate([scale, scale])
        cols = [c for c in range(32) if c not in (15, 31)]                 # F0 and its delta: discontinuous pick
        for n in ("max", "min", "range", "amean", "linregc2", "stddev"):
            j = NAMES.index(n)
            assert (np.abs(o[cols, j] - ref[cols, j]) <= 2e-5 * scale[cols]).all(), (k, n)
        j = NAMES.index("linregc1")
        assert (np.abs(o[cols, j] - ref[cols, j]) <= 2e-5 * scale[cols]).all(), (k, "linregc1")
        j = NAMES.index("linregerrQ")
        assert (np.abs(o[cols, j] - ref[cols, j]) <= 4e-5 * scale[cols] ** 2).all(), (k, "linregerrQ")
    b.close()


def test_functionals_mask_subsets_and_empty(hip, oracle):
    capi, ctx, plan = hip
    rng = np.random.default_rng(7)
    lens = [400 + 160 * 40, 0, 400]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    b = capi.Batch(plan, off)
    rows = np.diff(b.frame_offsets)
    assert list(rows) == [42, 0, 2]
    x = rng.standard_normal((int(rows.sum()), 32)).astype(np.float32)
    L = capi.load()
    for mask in (0x1ffff, 0x3f, 0xf00, 0x1f000, L.smilehip_functionals_is09_mask()):
        per = L.smilehip_functionals_count(mask)
        out = b.functionals_host(x, mask)
        assert out.shape == (3, 32 * per)
        ref0 = oracle.functionals(x[:39], mask).reshape(-1)
        ref2 = oracle.functionals(x[42:43], mask).reshape(-1)
        np.testing.assert_allclose(out[0], ref0, rtol=2e-6, atol=1e-12)
        np.testing.assert_allclose(out[2], ref2, rtol=2e-6, atol=1e-12)
        assert not out[1].any()
    assert L.smilehip_functionals_count(1 << 20) < 0
    b.close()
