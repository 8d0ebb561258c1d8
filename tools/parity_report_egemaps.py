#!/usr/bin/env python3
"""Parity statistics of the eGeMAPSv02 chain (SMILEHIP_CHAIN_EGEMAPS) against the golden file of the real binary
(tests/golden/egemaps_lld_synth.npz): per internal level, per LLD column and for the 88 functionals. Runs on a GPU box.
    python tools/parity_report_egemaps.py [out.json]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opensmile_amd import capi  # noqa: E402


def stats(a, b, scale=None):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if a.shape != b.shape:
        return {"shape_mismatch": [list(a.shape), list(b.shape)]}
    if a.size == 0:
        return {"n": 0}
    sc = np.maximum(np.abs(b).max(axis=0), 1e-6) if scale is None else scale
    e = np.abs(a - b) / sc
    bi = float((np.asarray(a, np.float32).view(np.uint32) == np.asarray(b, np.float32).view(np.uint32)).mean())
    return {"n": int(a.shape[0]), "bit_identical": bi, "frac_gt_1e-6": float((e > 1e-6).mean()), "max_scaled": float(e.max()), "p999": float(np.quantile(e, 0.999)), "p99": float(np.quantile(e, 0.99)),
            "frac_gt_1e-5": float((e > 1e-5).mean()), "frac_gt_1e-3": float((e > 1e-3).mean()),
            "worst_col": int(np.argmax(e.max(axis=0))) if e.ndim == 2 else 0}


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "egemaps_lld_synth.npz"))
    keys = sorted(k[4:] for k in g.files if k.startswith("pcm_"))
    ctx = capi.Context(0)
    plan = capi.Plan(ctx, capi.egemapsv02_config())
    pcms = [g["pcm_" + k] for k in keys]
    off = np.concatenate([[0], np.cumsum([len(p) for p in pcms])]).astype(np.int64)
    b = capi.Batch(plan, off)
    lld, func, t = b.run_host_egemaps(np.concatenate(pcms), taps=True)
    f20 = b.frame_offsets_frames(); f60 = t["frame_off60"]; fin = t["fin_off"]
    rep = {"cases": keys, "levels": {}, "lld_cols": {}, "func": {}}
    def cat(name):
        return np.concatenate([g[name + "_" + k].reshape(-1, g[name + "_" + k].shape[-1] if g[name + "_" + k].ndim == 2 else 1)
                               for k in keys if g[name + "_" + k].size], axis=0)
    raw = t["raw20"]
    rep["levels"]["loudness"] = stats(raw[:, 0:1], cat("loudness"))
    rep["levels"]["lspec"] = stats(raw[:, 1:5], cat("lspec"))
    rep["levels"]["flux"] = stats(raw[:, 5:6], cat("flux"))
    rep["levels"]["mfcc"] = stats(raw[:, 6:10], cat("mfcc"))
    rep["levels"]["energy2"] = stats(raw[:, 10:11], cat("energy2"))
    rep["levels"]["lpc"] = stats(t["lpc"][:, :11], cat("lpc"))
    rep["levels"]["formants"] = stats(t["formants"], cat("formants"))
    rep["levels"]["pitch"] = stats(t["pitch3"], cat("pitch"))
    jit = np.concatenate([t["jit4"][:, 0:1], t["shim_db"]], axis=1)
    rep["levels"]["jitter"] = stats(jit, cat("jitter"))
    rep["levels"]["harm"] = stats(t["harm6"], cat("harm"))
    # voiced-decision agreement and F0 error on commonly voiced frames
    pr = cat("pitch"); pg = t["pitch3"]
    both = (pr[:, 0] > 0) & (pg[:, 0] > 0)
    rep["levels"]["pitch_voicing_flip_frac"] = float(((pr[:, 0] > 0) != (pg[:, 0] > 0)).mean())
    rep["levels"]["pitch_f0_rel_max_on_common"] = float((np.abs(pg[both, 0] - pr[both, 0]) / pr[both, 0]).max()) if both.any() else 0.0
    rep["pending"] = t["pending"].tolist()
    ref_lld = np.concatenate([g["lld_" + k].reshape(-1, 25) for k in keys], axis=0)
    rep["lld"] = stats(lld, ref_lld)
    sc = np.maximum(np.abs(ref_lld).max(axis=0), 1e-6)
    for c in range(25):
        rep["lld_cols"][str(c)] = stats(lld[:, c:c + 1], ref_lld[:, c:c + 1], sc[c:c + 1])
    fr = []
    fg = []
    for i, k in enumerate(keys):
        r = g["func_" + k].reshape(-1, 88)
        if r.shape[0]:
            fr.append(r[0]); fg.append(func[i])
        else:
            assert not func[i].any(), "utterance without a 60 ms frame must give zeros"
    fr = np.array(fr); fg = np.array(fg)
    rel = np.abs(fg - fr) / np.maximum(np.abs(fr), 1e-2)
    rep["func"] = {"n_vectors": int(fr.shape[0]), "bit_identical_frac": float((fg.astype(np.float32).view(np.uint32) == fr.astype(np.float32).view(np.uint32)).mean()), "frac_within_1e-5": float((rel <= 1e-5).mean()),
                   "frac_within_1e-3": float((rel <= 1e-3).mean()), "max_rel": float(rel.max()), "p99_rel": float(np.quantile(rel, 0.99)),
                   "worst_cols": np.argsort(-rel.max(axis=0))[:8].tolist(), "worst_vals": np.sort(rel.max(axis=0))[::-1][:8].tolist()}
    rep["func"]["col87"] = [[float(a), float(b)] for a, b in zip(fg[:, 87], fr[:, 87])]
    rep["func"]["per_case_max_rel"] = {k: float(r) for k, r in zip([k for k in keys if g["func_" + k].size], rel.max(axis=1))}
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "egemaps_parity.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rep, open(out, "w"), indent=1)
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
