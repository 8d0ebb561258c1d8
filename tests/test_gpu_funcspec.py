"""General functionals (any cFunctionals instance, the nine families of ComParE_2016) through the C ABI's
smilehip_funcspec_matrix / smilehip_batch_funcspec against the CPU oracle (oracle/lld_oracle_funcspec.c, itself pinned
bit-exact against the real binary's functionals level in test_oracle_pin_funcspec.py).

The device walks every column in the reference's order, so all values agree bit for bit except the few that pass
through libm (log / exp: Means' nzgmean and flatness, the tanh soft limiter of ratio features): those are held to
1e-6 relative."""
import ctypes as C

import numpy as np
import pytest

from test_oracle_pin_f0 import KEYS_130

pytestmark = pytest.mark.gpu

INSTANCES = ["A", "B", "F0", "Nz", "LLD", "Delta"]
LIBM = {"flatness", "nzgmean", "peakMeanRel", "minMeanRel", "centroid", "linregc1", "qregc1", "qregc2", "stddevNorm",
        "covFallingSlope", "covRisingSlope"}


@pytest.fixture(scope="module")
def hip():
    from opensmile_amd import capi
    return capi, capi.Context(0)


def as_oracle_spec(oracle, spec):
    o = oracle.FuncSpec()
    assert C.sizeof(o) == C.sizeof(spec)
    C.memmove(C.byref(o), C.byref(spec), C.sizeof(spec))
    return o


def check(oracle, spec, dev, ref, what):
    names = oracle.funcspec_names(as_oracle_spec(oracle, spec))
    assert dev.shape == ref.shape == (ref.shape[0], len(names)), (what, dev.shape, ref.shape, len(names))
    for k, n in enumerate(names):
        d, r = dev[:, k], ref[:, k]
        if n in LIBM:
            err = np.abs(d.astype(np.float64) - r) / np.maximum(np.abs(r), 1e-6)
            assert err.max() <= 1e-6, f"{what}: {n} rel err {err.max():.3g}"
        else:
            same = d.view(np.uint32) == r.view(np.uint32)
            assert same.all(), f"{what}: {n} differs in columns {np.flatnonzero(~same)[:6]}: {d[~same][:3]} vs {r[~same][:3]}"


@pytest.mark.parametrize("inst", INSTANCES)
def test_funcspec_matrix_vs_oracle_on_real_llds(hip, oracle, golden_f0, inst):
    capi, ctx = hip
    spec = capi.funcspec_compare16(inst)
    for key in KEYS_130[:2]:
        x = golden_f0["lld130_" + key]
        if x.shape[0] < 8:
            continue
        cols = list(range(0, 12)) if inst in ("F0", "Nz") else list(range(6, 65)) + list(range(71, 130))
        xs = np.ascontiguousarray(x[:, cols])
        dev = capi.funcspec_matrix_host(ctx, spec, xs)
        ref = oracle.funcspec(xs, as_oracle_spec(oracle, spec))
        check(oracle, spec, dev, ref, f"{inst}/{key}")


@pytest.mark.parametrize("inst", INSTANCES)
def test_funcspec_edge_shapes(hip, oracle, inst):
    """Very short contours (every N from 1 to 9), constant and all-zero columns, NaN-free ratio limits."""
    capi, ctx = hip
    spec = capi.funcspec_compare16(inst)
    rng = np.random.default_rng(7)
    for rows in list(range(1, 10)) + [37]:
        x = rng.standard_normal((rows, 9)).astype(np.float32)
        x[:, 1] = 0.0                        # all zero (nonZeroFuncts: no value at all)
        x[:, 2] = 3.5                        # constant: range 0
        x[:, 3] = np.where(np.arange(rows) % 3 == 0, 0.0, x[:, 3])      # zeros interleaved
        x[:, 4] = np.abs(x[:, 4])
        x[:, 5] = np.round(x[:, 5] * 2) / 2  # ties
        dev = capi.funcspec_matrix_host(ctx, spec, x)
        ref = oracle.funcspec(x, as_oracle_spec(oracle, spec))
        assert np.isfinite(dev).all()
        check(oracle, spec, dev, ref, f"{inst}/rows{rows}")


def test_funcspec_percentile_sort_paths(hip, oracle):
    """The three percentile sorts -- one wave per contour up to 1024 rows (registers + lane exchanges), the workgroup sort in LDS
    up to 8192, global scratch beyond -- at their boundaries, with ties and signed zeros in the contour."""
    capi, ctx = hip
    spec = capi.funcspec_compare16("Nz")
    rng = np.random.default_rng(5)
    for rows in (63, 64, 65, 127, 129, 1000, 1024, 1025, 3000, 8192, 8193):
        x = rng.standard_normal((rows, 5)).astype(np.float32)
        x[:, 1] = np.round(x[:, 1] * 2) / 2          # ties
        x[::5, 2] = 0.0
        x[1::5, 2] = -0.0                            # signed zeros next to each other in sorted order
        x[:, 3] = np.abs(x[:, 3])
        dev = capi.funcspec_matrix_host(ctx, spec, x)
        ref = oracle.funcspec(x, as_oracle_spec(oracle, spec))
        check(oracle, spec, dev, ref, f"Nz/rows{rows}")


def test_funcspec_long_contour_global_sort(hip, oracle):
    """More rows than the LDS sort holds (8192): the percentile stage sorts in global scratch."""
    capi, ctx = hip
    spec = capi.funcspec_compare16("Nz")
    rng = np.random.default_rng(11)
    x = rng.standard_normal((20000, 3)).astype(np.float32)
    x[::7, 1] = 0.0
    dev = capi.funcspec_matrix_host(ctx, spec, x)
    ref = oracle.funcspec(x, as_oracle_spec(oracle, spec))
    check(oracle, spec, dev, ref, "Nz/20000 rows")


def test_funcspec_custom_spec_all_values(hip, oracle):
    """Every value of every family switched on, second / frame norms, nonX with a relative X, non-interpolated
    percentiles: the option paths the ComParE instances do not take."""
    capi, ctx = hip
    s = capi.FuncSpec()
    fams = [capi_f for capi_f in range(9)]
    s.n_fam = len(fams)
    for i, f in enumerate(fams):
        s.fam[i] = f
    s.period = 0.02
    s.ext_mask, s.ext_norm = 0xff, 1
    s.means_mask, s.means_norm = 0x1ffff, 1
    s.mom_mask, s.mom_stddev_norm, s.mom_ratio_limit = 0x3f, 1, 1
    s.reg_mask, s.reg_centroid_norm, s.reg_norm_coeff = 0x3ffff, 1, 1
    s.reg_norm_inputs, s.reg_centroid_abs, s.reg_centroid_limit, s.reg_ratio_limit, s.reg_old_buggy_qerr = 0, 0, 0, 0, 1
    s.pct_mask, s.pct_interp, s.n_pctl, s.n_range = 0x3f, 0, 3, 2
    s.pctl[0], s.pctl[1], s.pctl[2] = 0.05, 0.5, 0.95
    s.range_a[0], s.range_b[0], s.range_a[1], s.range_b[1] = 0, 2, 1, 0
    s.times_mask, s.times_norm, s.times_buggy_sec_norm = 0x1fff, 1, 0
    s.seg_mask, s.seg_norm, s.seg_algo, s.seg_max_num = 0x1f, 2, 1, 5
    s.seg_min_lng, s.seg_auto_min_lng, s.seg_pause_min_lng, s.seg_x_is_rel, s.seg_x = 2, 0, 1, 1, 0.0
    s.lpc_gain, s.lpc_coeffs, s.lpc_first, s.lpc_order = 1, 1, 1, 8
    s.pk_mask, s.pk_norm, s.pk_ratio_limit, s.pk_dyn_rel, s.pk_rel_thresh = 0xffffffff, 0, 0, 1, 0.3
    rng = np.random.default_rng(3)
    x = np.abs(rng.standard_normal((400, 7))).astype(np.float32)
    x[rng.random((400, 7)) < 0.3] = 0.0                  # runs of the column minimum: nonX with a relative X = 0
    dev = capi.funcspec_matrix_host(ctx, s, x)
    ref = oracle.funcspec(x, as_oracle_spec(oracle, s))
    names = oracle.funcspec_names(as_oracle_spec(oracle, s))
    assert dev.shape == ref.shape
    err = np.abs(dev.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-6)
    assert err.max() <= 1e-6, [(names[k], float(err[:, k].max())) for k in np.argsort(-err.max(axis=0))[:4]]
    exact = (dev.view(np.uint32) == ref.view(np.uint32)).mean()
    assert exact > 0.97, exact


@pytest.mark.parametrize("algo", [1, 2])
def test_funcspec_segments_nonx_eqx_seconds(hip, oracle, algo):
    """Segments nonX / eqX with second norm and numSegments, as the GeMAPS sets configure them, on a voiced/unvoiced
    pattern (runs of zeros)."""
    capi, ctx = hip
    s = capi.FuncSpec()
    s.n_fam, s.period = 1, 0.01
    s.fam[0] = 6
    s.seg_mask, s.seg_norm, s.seg_algo, s.seg_max_num = 0x1f, 1, algo, 1000
    s.seg_min_lng, s.seg_auto_min_lng, s.seg_pause_min_lng, s.seg_x = 3, 1, 2, 0.0
    rng = np.random.default_rng(21)
    x = rng.standard_normal((600, 8)).astype(np.float32)
    for c in range(8):
        pos = 0
        while pos < 600:
            run = int(rng.integers(1, 40))
            if rng.random() < 0.5:
                x[pos:pos + run, c] = 0.0
            pos += run
    dev = capi.funcspec_matrix_host(ctx, s, x)
    ref = oracle.funcspec(x, as_oracle_spec(oracle, s))
    assert np.array_equal(dev.view(np.uint32), ref.view(np.uint32))
    assert (ref[:, 0] > 0).all()


SEG_ALGOS = dict(relTh=0, nonX=1, eqX=2, mrelTh=3, absTh=4, NArelTh=5, NAmrelTh=6, NAabsTh=7, delta=8, delt2=9, chX=10)


@pytest.mark.parametrize("algo", sorted(SEG_ALGOS))
def test_funcspec_every_segmentation_algorithm(hip, oracle, algo):
    """Every segmentationAlgorithm of functionalSegments.cpp:118-155 (the oracle's restatement is pinned on the binary in
    test_oracle_pin_is10.py::test_segments_every_algorithm_bit_exact): contours with plateaus at 0 and at their minimum, long and
    short, automatic and explicit minimum lengths, all three time norms, more segments than maxNumSeg."""
    capi, ctx = hip
    rng = np.random.default_rng(SEG_ALGOS[algo] + 5)
    for n_rows, norm, max_num, min_lng, ravg in ((600, 1, 100, None, 0), (77, 2, 4, 2, 3), (1500, 0, 20, None, 0), (9, 2, 20, 1, 0)):
        s = capi.FuncSpec()
        s.n_fam, s.period = 1, 0.01
        s.fam[0] = 6
        s.seg_mask, s.seg_norm, s.seg_algo, s.seg_max_num = 0x1f, norm, SEG_ALGOS[algo], max_num
        s.seg_min_lng, s.seg_auto_min_lng = (3, 1) if min_lng is None else (min_lng, 0)
        s.seg_pause_min_lng, s.seg_x, s.seg_x_is_rel = 2, 0.0, int(n_rows == 1500)
        s.seg_n_thresholds = 3
        th = (0.2, 0.5, 0.85) if algo in ("relTh", "NArelTh") else (0.6, 1.0, 1.7) if "mrel" in algo else (-0.4, 0.1, 0.9)
        for j in range(3):
            s.seg_thresholds[j] = th[j]
        s.seg_ravg_lng, s.seg_range_rel_threshold = ravg, 0.15
        x = np.cumsum(rng.standard_normal((n_rows, 8)), axis=0).astype(np.float32) * 0.3
        x[:, 4:] = np.abs(x[:, 4:])
        for c in range(8):
            pos = 0
            while pos < n_rows:
                run = int(rng.integers(1, 30))
                if rng.random() < 0.4:
                    x[pos:pos + run, c] = 0.0 if c < 6 else x[:, c].min()
                pos += run
        dev = capi.funcspec_matrix_host(ctx, s, x)
        ref = oracle.funcspec(x, as_oracle_spec(oracle, s))
        assert np.array_equal(dev.view(np.uint32), ref.view(np.uint32)), (algo, n_rows, dev[:2], ref[:2])
        if algo != "absTh" and n_rows >= 77:
            assert (ref[:, 0] > 0).any(), (algo, n_rows)


def test_funcspec_percentile_quotients_and_level_times(hip, oracle):
    """Percentiles.pctlquotient[] and Times.upleveltime[] / downleveltime[] (the oracle's restatement is pinned on the binary in
    test_oracle_pin_is10.py::test_percentile_quotients_and_level_times_bit_exact): the pinned specs on contours with zero plateaus
    (zero numerators and denominators), short and long (the global sort path), both percentile read-outs."""
    capi, ctx = hip
    from test_oracle_pin_is10 import quot_time_cases
    rng = np.random.default_rng(77)
    for n_rows in (5, 300, 2500):
        x = np.abs(np.cumsum(rng.standard_normal((n_rows, 6)), axis=0)).astype(np.float32)
        x[: n_rows // 3, :3] = 0.0
        x[:, 5] = -x[:, 5]
        for k, so in quot_time_cases().items():
            s = capi.FuncSpec()
            assert C.sizeof(s) == C.sizeof(so)
            C.memmove(C.byref(s), C.byref(so), C.sizeof(s))
            dev = capi.funcspec_matrix_host(ctx, s, x)
            ref = oracle.funcspec(x, so)
            names = oracle.funcspec_names(so)
            assert dev.shape == ref.shape == (6, len(names)), (k, dev.shape, ref.shape, len(names))
            for c, nm in enumerate(names):
                if nm.startswith("pctlquotient"):                       # the soft limiter goes through exp
                    err = np.abs(dev[:, c].astype(np.float64) - ref[:, c]) / np.maximum(np.abs(ref[:, c]), 1e-6)
                    assert err.max() <= 1e-6, (k, nm, dev[:, c], ref[:, c])
                else:
                    assert np.array_equal(dev[:, c].view(np.uint32), ref[:, c].view(np.uint32)), (k, n_rows, nm, dev[:, c], ref[:, c])


def test_funcspec_modulation_spectrum(hip, oracle):
    """The Modulation family (cFunctionalModulation's ModulationSpec values; the oracle's restatement is pinned on the binary in
    test_oracle_pin_is10.py::test_modulation_spectrum_bit_exact): the three pinned option sets on contours of 40 .. 2500 rows (one
    short window zero-padded to 64 / 128 / 512 points, several windows with a dropped tail), with and without nonZeroFuncts, next
    to another family in the same instance. Every value the oracle's bits; a contour too short for the 33-value limit gives NaNs
    on both sides."""
    capi, ctx = hip
    from test_oracle_pin_is10 import modulation_cases
    rng = np.random.default_rng(31)
    for n_rows in (40, 98, 300, 998, 2500, 20):
        x = np.abs(np.cumsum(rng.standard_normal((n_rows, 5)), axis=0)).astype(np.float32)
        x[n_rows // 4: n_rows // 3, :2] = 0.0
        for k, mc in modulation_cases().items():
            for nz in (0, 1):
                s = capi.FuncSpec()
                s.n_fam, s.period, s.non_zero_functs = 2, 0.01, nz
                s.fam[0], s.fam[1] = 14, 0
                s.ext_mask, s.ext_norm = 0x07, 2
                s.mod_win_frames, s.mod_step_frames, s.mod_n_bins = mc.win_frames, mc.step_frames, mc.n_bins
                s.mod_win_func, s.mod_remove_nz_mean = mc.win_func, mc.remove_nz_mean
                s.mod_min_freq, s.mod_max_freq = mc.min_freq, mc.max_freq
                dev = capi.funcspec_matrix_host(ctx, s, x)
                ref = oracle.funcspec(x, as_oracle_spec(oracle, s))
                assert dev.shape == ref.shape == (5, mc.n_bins + 3)
                assert np.array_equal(dev.view(np.uint32), ref.view(np.uint32)) or \
                    (np.isnan(ref[:, :mc.n_bins]).all() and np.isnan(dev[:, :mc.n_bins]).all() and
                     np.array_equal(dev[:, mc.n_bins:].view(np.uint32), ref[:, mc.n_bins:].view(np.uint32))), (k, n_rows, nz, dev[0, :4], ref[0, :4])
                if n_rows >= 40 and not nz:
                    assert np.isfinite(ref).all(), (k, n_rows)


def test_batch_funcspec_ragged_with_cut_and_extra_row(hip, oracle):
    """smilehip_batch_funcspec on a ragged batch: per-utterance rows = max(1, rows - cut) (+ one extra row), column
    sub-ranges, utterances without rows."""
    capi, ctx = hip
    from opensmile_amd import synth
    plan = capi.Plan(ctx, capi.compare16_config())
    lens = [48000, 0, 9000, 1760, 24000, 1600]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pcm = np.concatenate([synth.utterance(70 + i, n) if n else np.zeros(0, np.int16) for i, n in enumerate(lens)])
    b = capi.Batch(plan, off)
    lld = b.run_host(pcm)
    rng = np.random.default_rng(5)
    for inst, c0, nc, cut, with_extra in (("A", 6, 4, 3, False), ("B", 10, 55, 0, True), ("Nz", 0, 6, 5, False),
                                          ("LLD", 6, 59, 1, False), ("Delta", 71, 59, 3, False), ("F0", 0, 1, 3, False)):
        spec = capi.funcspec_compare16(inst)
        extra = rng.standard_normal((len(lens), nc)).astype(np.float32) if with_extra else None
        dev = b.funcspec_host(lld, spec, c0, nc, cut, extra)
        per = capi.funcspec_count(spec)
        assert dev.shape == (len(lens), nc * per)
        for u in range(len(lens)):
            x = lld[b.frame_offsets[u]:b.frame_offsets[u + 1], c0:c0 + nc]
            if x.shape[0] == 0:
                assert not dev[u].any()
                continue
            x = x[:max(1, x.shape[0] - cut)]
            if with_extra:
                x = np.concatenate([x, extra[u:u + 1]], axis=0)
            ref = oracle.funcspec(np.ascontiguousarray(x), as_oracle_spec(oracle, spec))
            check(oracle, spec, dev[u].reshape(nc, per), ref, f"{inst}/utt{u}")
    b.close()


def test_funcspec_rejects_unusable_specs(hip):
    capi, ctx = hip
    s = capi.funcspec_compare16("A")
    s.lpc_order = 9
    with pytest.raises(capi.SmileHipError, match="Lpc.order"):
        capi.funcspec_count(s)
    s = capi.funcspec_compare16("A")
    s.seg_algo = 11
    with pytest.raises(capi.SmileHipError, match="segmentationAlgorithm"):
        capi.funcspec_count(s)
    s = capi.funcspec_compare16("Nz")
    s.range_b[0] = 5
    with pytest.raises(capi.SmileHipError, match="pctlrange"):
        capi.funcspec_count(s)
    with pytest.raises(capi.SmileHipError, match="unknown ComParE_2016 functionals instance"):
        capi.funcspec_compare16("C")
