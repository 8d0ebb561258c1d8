"""config/plp/PLP_0_D_A.conf on the GPU (R8's IDFT / Durbin / cepstrum branch after the MFCC front end):
6 PLP cepstra + delta + accel through smilehip_lld_run, fast and reference-order kernels, against golden
outputs of the real reference binary and against the CPU oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ["u2_16000", "u3_16000", "u10_16000", "u1_16000", "u0_16000", "u7_399", "u7_400", "u7_560", "u7_1000", "u5_160000"]


@pytest.fixture(scope="module", params=["fast", "generic"])
def hip(request):
    os.environ["SMILEHIP_FORCE_GENERIC"] = "1" if request.param == "generic" else "0"
    from opensmile_amd import capi
    ctx = capi.Context(0)
    plan = capi.Plan(ctx, capi.plp_0_d_a_config())
    os.environ.pop("SMILEHIP_FORCE_GENERIC", None)
    assert (plan.geometry.n_static, plan.geometry.n_out) == (6, 18)
    return capi, ctx, plan, request.param


def check(out, ref, what):
    assert out.shape == ref.shape, f"{what}: {out.shape} vs {ref.shape}"
    if not ref.size:
        return
    assert np.isfinite(out).all()
    # per-frame scale of the static block (c0 dominates), deltas on the same scale
    scale = np.abs(ref[:, :6]).max(axis=1, keepdims=True)
    err = (np.abs(out - ref) / np.maximum(scale, 1e-30)).max()
    assert err <= 1e-5, f"{what}: {err:.2e}"


def test_plp_golden_batch_ragged(hip, golden_plp):
    capi, ctx, plan, kind = hip
    pcms = [golden_plp["pcm_" + k] for k in KEYS]
    refs = [golden_plp["out_" + k] for k in KEYS]
    off = np.concatenate([[0], np.cumsum([len(p) for p in pcms])]).astype(np.int64)
    b = capi.Batch(plan, off)
    np.testing.assert_array_equal(np.diff(b.frame_offsets), [r.shape[0] for r in refs])
    out = b.run_host(np.concatenate(pcms))
    for i, k in enumerate(KEYS):
        check(out[b.frame_offsets[i]:b.frame_offsets[i + 1]], refs[i], f"{kind} {k}")
    b.close()


def test_plp_vs_oracle_and_delta_tail(hip, oracle):
    capi, ctx, plan, kind = hip
    from opensmile_amd import synth
    pcm = synth.utterance(21, 160000)
    b = capi.Batch(plan, np.array([0, 160000], np.int64))
    out = b.run_host(pcm)
    check(out, oracle.plp_chain(pcm), f"{kind} 10 s")
    # R13 given the GPU's own static block: bit-exact (same float expressions as the reference)
    de = oracle.delta_chain(out[:, :6].copy(), 2, 2)
    assert np.array_equal(out[:, 6:12], de[0]) and np.array_equal(out[:, 12:18], de[1])
    b.close()
