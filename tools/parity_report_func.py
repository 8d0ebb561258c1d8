#!/usr/bin/env python3
"""Parity statistics of the functionals level of ComParE_2016 (6373 values per utterance): GPU (whole-level chain +
smilehip_batch_functionals_compare16) against the REAL reference binary (oracle/_ref/SMILExtract -htkoutput) over fresh
synthetic utterances. Prints one JSON object: per functional (value-name suffix) the median / 99th-percentile error
relative to max(|reference|, 1e-2) and the share of values within 1e-3 and bit-identical; and the attribution of that
residue: the drift of every LLD column at the LLD level (GPU vs the binary's lld file, relative to the column's scale), the
functionals error grouped by the LLD column they summarise, and per functional the LLD columns that carry its tail."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=24)
    ap.add_argument("--first", type=int, default=300)
    args = ap.parse_args()
    from opensmile_amd import capi, synth
    from oracle import lldo
    ctx = capi.Context(0)
    plan = capi.Plan(ctx, capi.compare16_config())
    lens = [160000 if i % 3 else 48000 + 1600 * i for i in range(args.utts)]
    pcms = [synth.utterance(args.first + i, n) for i, n in enumerate(lens)]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    b = capi.Batch(plan, off)
    lld, func = b.run_host_with_functionals16(np.concatenate(pcms))[:2]
    row_off = b.frame_offsets
    names = list(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                      "compare16_func_synth.npz"))["names"])
    suffix = np.array([str(n).rsplit("_", 1)[1] for n in names])
    refs = [lldo.run_reference_func("compare16/ComParE_2016.conf", p) for p in pcms]
    ref = np.stack([r[0][0] for r in refs])
    prefix = np.array([str(n).rsplit("_", 1)[0] for n in names])
    # LLD-level drift per column: largest |GPU - binary| over all rows, relative to the column's largest magnitude
    ref_lld = np.concatenate([r[1] for r in refs])
    assert ref_lld.shape == lld.shape, (ref_lld.shape, lld.shape, row_off[-1])
    col_scale = np.maximum(np.abs(ref_lld).max(axis=0), 1e-12)
    lld_drift = np.abs(lld.astype(np.float64) - ref_lld).max(axis=0) / col_scale
    lld_rows_off = (np.abs(lld.astype(np.float64) - ref_lld) > 1e-5 * col_scale[None, :]).mean(axis=0)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lld_names = open(os.path.join(root, "tests", "golden", "files", "compare16_lld_u3.csv")).readline().strip().split(";")[2:]
    assert len(lld_names) == lld.shape[1]
    err = np.abs(func.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-2)
    same = func.view(np.uint32) == ref.view(np.uint32)
    res = {"utterances": args.utts, "values": int(err.size), "reference": "oracle/_ref/SMILExtract -C compare16/ComParE_2016.conf -htkoutput",
           "all": {"median": float(np.median(err)), "p99": float(np.quantile(err, 0.99)), "within_1e-3": float((err <= 1e-3).mean()),
                   "bit_identical": float(same.mean())}, "by_functional": {}}
    for sfx in sorted(set(suffix), key=lambda x: -float(np.quantile(err[:, suffix == x], 0.99))):
        e = err[:, suffix == sfx]
        res["by_functional"][sfx] = {"n": int(e.size), "median": float(np.median(e)), "p99": float(np.quantile(e, 0.99)),
                                     "within_1e-3": float((e <= 1e-3).mean()), "bit_identical": float(same[:, suffix == sfx].mean())}
    # functionals grouped by the LLD column they summarise (the value name is <lld column>_<functional>)
    res["by_lld_column"] = {}
    for pf in sorted(set(prefix), key=lambda x: -float(np.quantile(err[:, prefix == x], 0.99))):
        e = err[:, prefix == pf]
        ent = {"n": int(e.size), "p99": float(np.quantile(e, 0.99)), "within_1e-3": float((e <= 1e-3).mean()),
               "bit_identical": float(same[:, prefix == pf].mean())}
        if lld_names is not None and pf in lld_names:
            c = lld_names.index(pf)
            ent["lld_drift_max"] = float(lld_drift[c])
            ent["lld_rows_beyond_1e-5"] = float(lld_rows_off[c])
        res["by_lld_column"][pf] = ent
    # per functional: the three LLD columns that carry its tail
    for sfx in res["by_functional"]:
        m = suffix == sfx
        worst = sorted(set(prefix[m]), key=lambda x: -float(np.quantile(err[:, m & (prefix == x)], 0.99)))[:3]
        res["by_functional"][sfx]["worst_lld_columns"] = [
            {"lld": w, "p99": float(np.quantile(err[:, m & (prefix == w)], 0.99)),
             "lld_drift_max": (float(lld_drift[lld_names.index(w)]) if lld_names is not None and w in lld_names else None)} for w in worst]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
