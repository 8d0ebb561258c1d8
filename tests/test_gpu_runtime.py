"""Runtime behaviour of the C ABI on the device: no leaks over create/destroy cycles, non-default
streams, several batches of one plan in flight, error paths that must fail loudly."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_no_device_memory_leak_over_plan_and_batch_cycles():
    import torch
    from opensmile_amd import capi, synth
    ctx = capi.Context(0)
    pcm = np.concatenate([synth.utterance(3, 16000), synth.utterance(4, 8000)])
    off = np.array([0, 16000, 24000], np.int64)

    def cycle(cfg_fn):
        plan = capi.Plan(ctx, cfg_fn())
        b = capi.Batch(plan, off)
        out = b.run_host(pcm)
        if cfg_fn is capi.is09_lld_config:
            b.functionals_host(out)
        b.close()
        plan.close()
        return out

    for fn in (capi.mfcc12_0_d_a_config, capi.plp_0_d_a_config, capi.is09_lld_config, capi.compare16_ab_config,
               capi.compare16_f0_config, capi.compare16_config):
        cycle(fn)                                        # warm allocator pools
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(40):
        for fn in (capi.mfcc12_0_d_a_config, capi.plp_0_d_a_config, capi.is09_lld_config, capi.compare16_ab_config,
                   capi.compare16_f0_config, capi.compare16_config):
            cycle(fn)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 * 2 ** 20, f"device memory shrank by {(free0 - free1) / 2 ** 20:.1f} MiB over 240 cycles"


def test_two_batches_on_two_streams_match_serial_results():
    import torch
    from opensmile_amd import capi, synth
    ctx = capi.Context(0)
    plan = capi.Plan(ctx)
    S = 160000
    pcm_a, off_a = synth.corpus_tiled(40, S, n_unique=8)
    pcm_b = np.concatenate([synth.utterance(50 + i, 48000) for i in range(30)])
    off_b = np.arange(31, dtype=np.int64) * 48000
    ba, bb = capi.Batch(plan, off_a), capi.Batch(plan, off_b)
    ref_a, ref_b = ba.run_host(pcm_a), bb.run_host(pcm_b)
    da, db = torch.from_numpy(pcm_a).cuda(), torch.from_numpy(pcm_b).cuda()
    oa = torch.zeros((ba.total_rows, 39), device="cuda")
    ob = torch.zeros((bb.total_rows, 39), device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(5):
        ba.run_device(da.data_ptr(), oa.data_ptr(), 39, s1.cuda_stream)
        bb.run_device(db.data_ptr(), ob.data_ptr(), 39, s2.cuda_stream)
    s1.synchronize()
    s2.synchronize()
    assert np.array_equal(oa.cpu().numpy(), ref_a) and np.array_equal(ob.cpu().numpy(), ref_b)
    ba.close()
    bb.close()


def _rate(cfg, fs):
    cfg.sample_rate = fs
    return cfg


def test_error_paths_fail_loudly():
    from opensmile_amd import capi
    L = capi.load()
    ctx = capi.Context(0)
    with pytest.raises(capi.SmileHipError):
        capi.Context(999)                                # no such device
    cfg = capi.mfcc12_0_d_a_config()
    cfg.n_bands = 0
    with pytest.raises(capi.SmileHipError):
        capi.Plan(ctx, cfg)
    cfg = capi.is09_lld_config()
    cfg.sma_win = 4                                      # even smaWin is not a cContourSmoother window
    with pytest.raises(capi.SmileHipError):
        capi.Plan(ctx, cfg)
    cfg = capi.compare16_f0_config()
    cfg.sample_rate = 96000.0                            # 60 ms -> 8192-point spectrum: the F0 kernels are instantiated up to 4096 (48 kHz)
    with pytest.raises(capi.SmileHipError, match="512 .. 4096"):
        capi.Plan(ctx, cfg)
    cfg = capi.compare16_config()
    cfg.sample_rate = 96000.0                            # (the 20 ms part refuses first: 2048 points)
    with pytest.raises(capi.SmileHipError):
        capi.Plan(ctx, cfg)
    cfg = capi.egemapsv02_config()
    cfg.sample_rate = 11025.0                            # cSpecResample would give 221 samples per 20 ms frame: not the kernel's 220
    with pytest.raises(capi.SmileHipError):
        capi.Plan(ctx, cfg)
    capi.Plan(ctx, _rate(capi.compare16_config(), 44100.0)).close()     # the rates of tests/test_gpu_rates.py build
    plan = capi.Plan(ctx)
    with pytest.raises(capi.SmileHipError):
        capi.Batch(plan, np.array([0, 100, 50], np.int64))        # offsets must be non-decreasing
    b = capi.Batch(plan, np.array([0, 16000], np.int64))
    with pytest.raises(capi.SmileHipError):
        b.run_device(0, 0, 39)                           # null device pointers
    with pytest.raises(capi.SmileHipError):
        b.func_rows()                                    # functionals belong to IS09 plans
    other = capi.Plan(ctx)
    rc = L.smilehip_lld_run(other._h, b._h, C.c_void_p(8), C.c_void_p(8), 39, None)
    assert rc != 0 and b"mismatch" in L.smilehip_last_error()
    host_only = capi.Plan(None)
    with pytest.raises(capi.SmileHipError):
        capi.Batch(host_only, np.array([0, 16000], np.int64)).run_host(np.zeros(16000, np.int16))
    b.close()
