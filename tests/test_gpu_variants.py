"""The other six configs of config/mfcc and config/plp on the GPU (chains MFCC / PLP with append_log_energy and/or cms):
golden outputs of the real binary, ragged batches against the oracle, the energy column bit-exact."""
import numpy as np
import pytest

from test_oracle_pin_variants import KEYS, NAMES, variant_tolerance

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden_variants():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "htk_variants_synth.npz"))


@pytest.fixture(scope="module")
def ctx():
    from opensmile_amd import capi
    return capi, capi.Context(0)


@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("name", NAMES)
def test_variant_golden_batch(ctx, oracle, golden_variants, name, generic, monkeypatch):
    capi, c = ctx
    if generic:
        monkeypatch.setenv("SMILEHIP_FORCE_GENERIC", "1")      # the reference-order frame kernel
    plan = capi.Plan(c, capi.htk_variant_config(name))
    _, plp, energy, _ = oracle.HTK_VARIANTS[name]
    n_cep = (5 if plp else 12) + (0 if energy else 1)
    assert plan.geometry.n_static == n_cep + (1 if energy else 0) and plan.geometry.n_out == 3 * plan.geometry.n_static
    pcms = [golden_variants["pcm_" + k] for k in KEYS]
    off = np.concatenate([[0], np.cumsum([len(p) for p in pcms])]).astype(np.int64)
    b = capi.Batch(plan, off)
    out = b.run_host(np.concatenate(pcms))
    for i, k in enumerate(KEYS):
        o = out[b.frame_offsets[i]:b.frame_offsets[i + 1]]
        ref = golden_variants[name + "_" + k]
        raw = oracle.htk_variant_chain(name[:-2], pcms[i]) if name.endswith("_Z") else None
        variant_tolerance(o, ref, n_cep, f"{name} {k} generic={generic}", raw)
        if energy:                                              # the energy column follows the reference's own summation order
            D = n_cep + 1
            assert np.array_equal(o[:, n_cep].view(np.uint32), ref[:, n_cep].view(np.uint32)), (name, k)
    b.close()
    plan.close()


@pytest.mark.parametrize("name", ["MFCC12_E_D_A_Z", "PLP_E_D_A_Z"])
def test_variant_vs_oracle_ragged(ctx, oracle, name):
    capi, c = ctx
    from opensmile_amd import synth
    plan = capi.Plan(c, capi.htk_variant_config(name))
    _, plp, energy, _ = oracle.HTK_VARIANTS[name]
    n_cep = (5 if plp else 12) + (0 if energy else 1)
    lens = [160000, 399, 400, 401, 560, 0, 1200, 5280, 5281, 48000, 160001]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pcm = np.concatenate([synth.utterance(20 + i, n) if n else np.zeros(0, np.int16) for i, n in enumerate(lens)])
    b = capi.Batch(plan, off)
    out = b.run_host(pcm)
    oracle.use_reference_fft(False)
    for i, n in enumerate(lens):
        ref = oracle.htk_variant_chain(name, pcm[off[i]:off[i + 1]])
        raw = oracle.htk_variant_chain(name[:-2], pcm[off[i]:off[i + 1]])
        o = out[b.frame_offsets[i]:b.frame_offsets[i + 1]]
        variant_tolerance(o, ref, n_cep, f"{name} len{n}", raw)
    b.close()
    plan.close()


def test_variant_no_deltas_in_place(ctx, oracle):
    """n_delta = 0: the frame kernel writes the output rows directly; energy column and mean subtraction work in place."""
    capi, c = ctx
    from opensmile_amd import synth
    cfg = capi.htk_variant_config("MFCC12_E_D_A_Z")
    cfg.n_delta = 0
    plan = capi.Plan(c, cfg)
    pcm = synth.utterance(5, 32000)
    b = capi.Batch(plan, np.array([0, 32000], np.int64))
    out = b.run_host(pcm)
    oracle.use_reference_fft(False)
    ref = oracle.htk_variant_chain("MFCC12_E_D_A_Z", pcm)[:, :13]
    assert out.shape == ref.shape
    scale = np.abs(oracle.htk_variant_chain("MFCC12_E_D_A", pcm)[:, :12]).max(axis=1, keepdims=True)
    assert (np.abs(out[:, :12] - ref[:, :12]) / scale).max() <= 1e-5
    assert np.array_equal(out[:, 12], ref[:, 12])
    b.close()
    plan.close()


def test_unknown_variant_name_is_refused(ctx):
    capi, c = ctx
    with pytest.raises(capi.SmileHipError):
        capi.htk_variant_config("MFCC13_0_D_A")
