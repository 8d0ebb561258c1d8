"""Accuracy gates of SURVEY.md §8(d) / BASELINE.json north_star (<= 1e-5 relative).

Element-wise pure relative error is not a usable gate (a float64 restatement
already differs from the float32 reference by 4e-3 on near-zero cepstra), so
"relative" is normalised by a scale:
  (ii)  max_t max_i |d[t,i]| / max_i |ref[t,i]|          per-frame-scaled error
  (iii) |d| <= RTOL * max(|ref|, s_col), s_col = 99th percentile of |ref| per column

Delta / acceleration columns are a fixed linear stencil over the static block
(weights i/(2 sum i^2) <= 0.2, R13), so their error is judged on the scale of
the STATIC column they derive from (`block` = number of static columns): a
stationary signal has deltas ~0 whose own magnitude is not a meaningful scale.
"""
import numpy as np

RTOL = 1e-5


def assert_bits_equal(out, ref, what=""):
    """Round 3: every reference-order chain (everything but the fast MFCC / PLP kernel, which keeps its own transform and
    the RTOL gate below) follows the reference's operation order stage by stage -- rdft network, glibc's logf / expf /
    log10f, FLOAT_DMEM accumulators -- so its outputs are compared with the real binary's (and the oracle's) as bit patterns."""
    o = np.ascontiguousarray(out, np.float32)
    r = np.ascontiguousarray(ref, np.float32)
    assert o.shape == r.shape, f"{what}: {o.shape} vs {r.shape}"
    same = o.view(np.uint32) == r.view(np.uint32)
    record("assert_bits_equal", what=what, cells=same.size, identical=float(same.mean()) if same.size else 1.0)
    assert same.all(), (f"{what}: {int((~same).sum())} of {same.size} cells differ from the reference's bits, first at "
                        f"{tuple(np.argwhere(~same)[0])}, largest |difference| {np.abs(o.astype(np.float64) - r).max():.3g}")


def record(gate, **measured):
    """Margin bookkeeping: with SMILEHIP_GATE_LOG=<file> every gate appends what it measured (one JSON object per line), so
    that the thresholds can be kept at about twice the measured error (profiles/rNN_gate_margins.json) instead of drifting
    apart from it."""
    import json
    import os
    path = os.environ.get("SMILEHIP_GATE_LOG")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps({"gate": gate, **{k: (float(v) if not isinstance(v, str) else v) for k, v in measured.items()}}) + "\n")


def corpus_col_scale(refs, block):
    """s_col of gate (iii): 99th percentile of |ref| per STATIC column, pooled
    over the evaluation corpus `refs` (list of T x D reference matrices)."""
    pool = np.concatenate([np.abs(np.asarray(r, np.float64))[:, :block] for r in refs if len(r)])
    return np.percentile(pool, 99, axis=0)


def _static_scales(ref, block, col_scale=None):
    ref = np.abs(np.asarray(ref, np.float64))
    stat = ref[:, :block]
    frame_scale = stat.max(axis=1)                      # per frame
    if col_scale is None:
        col_scale = np.percentile(stat, 99, axis=0)     # per static column (single-array corpus)
    reps = ref.shape[1] // block
    return frame_scale, np.tile(col_scale, reps)


def frame_scaled_err(out, ref, block=None):
    out = np.asarray(out, np.float64)
    ref = np.asarray(ref, np.float64)
    if ref.size == 0:
        return 0.0
    block = block or ref.shape[1]
    frame_scale, _ = _static_scales(ref, block)
    d = np.abs(out - ref).max(axis=1)
    nz = frame_scale > 0
    if (~nz).any():
        assert d[~nz].max() == 0.0, "non-zero error where the reference frame is all zero"
    return float((d[nz] / frame_scale[nz]).max()) if nz.any() else 0.0


def column_scaled_err(out, ref, block=None, col_scale=None):
    out = np.asarray(out, np.float64)
    ref = np.asarray(ref, np.float64)
    if ref.size == 0:
        return 0.0
    block = block or ref.shape[1]
    _, col_scale = _static_scales(ref, block, col_scale)
    scale = np.maximum(np.abs(ref), col_scale[None, :])
    d = np.abs(out - ref)
    z = scale == 0
    if z.any():
        assert d[z].max() == 0.0, "non-zero error where the reference column is all zero"
    return float((d[~z] / scale[~z]).max()) if (~z).any() else 0.0


def column_scaled_pass_rate(out, ref, block=None, rtol=RTOL, col_scale=None):
    out = np.asarray(out, np.float64)
    ref = np.asarray(ref, np.float64)
    if ref.size == 0:
        return 1.0
    block = block or ref.shape[1]
    _, col_scale = _static_scales(ref, block, col_scale)
    scale = np.maximum(np.abs(ref), col_scale[None, :])
    return float((np.abs(out - ref) <= rtol * scale).mean())


def assert_parity(out, ref, block=None, rtol=RTOL, what="", col_scale=None):
    """Gate (ii): per-frame-scaled error <= rtol, always. Every coefficient of
    a frame is a fixed linear map (DCT x lifter) of that frame's log-mel vector,
    so the frame's largest coefficient is the scale any float32 implementation's
    round-off lives on.
    Gate (iii): |d| <= rtol * max(|ref|, s_col) for every element when a
    corpus-level `col_scale` (corpus_col_scale) is given. Without one, s_col
    falls back to the percentile of this array alone, which is degenerate for a
    stationary signal (the corpus contract's 100 Hz square wave has columns that
    are small constants; float32 rounding of the ~20.0-valued log-mel inputs
    alone puts ~2e-5 absolute on every cepstral coefficient), so the fallback
    only requires pass-rate >= 99 % and worst element <= 10 rtol."""
    assert out.shape == ref.shape, f"{what}: shape {out.shape} != {ref.shape}"
    assert np.isfinite(out).all(), f"{what}: non-finite output"
    e2 = frame_scaled_err(out, ref, block)
    assert e2 <= rtol, f"{what}: per-frame-scaled error {e2:.3e} > {rtol:.0e}"
    e3 = column_scaled_err(out, ref, block, col_scale)
    if col_scale is not None:
        assert e3 <= rtol, f"{what}: column-scaled error {e3:.3e} > {rtol:.0e}"
        return e2, e3
    rate = column_scaled_pass_rate(out, ref, block, rtol)
    if ref.shape[0] < 50:      # too few frames for a percentile / pass-rate to mean anything
        rate = 1.0
    assert rate >= 0.99 and e3 <= 10 * rtol, \
        f"{what}: column-scaled error {e3:.3e}, pass-rate {rate:.4f}"
    return e2, e3
