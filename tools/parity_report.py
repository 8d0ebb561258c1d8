#!/usr/bin/env python3
"""Parity statistics of the whole ComParE_2016 LLD level: GPU (chain COMPARE) against the REAL reference binary
(oracle/_ref/SMILExtract, -lldhtkoutput) over fresh synthetic utterances. Prints one JSON object: per column group the
largest deviation relative to the column's scale and the share of rows deviating by more than 1e-5 / 1e-3."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=40)
    ap.add_argument("--first", type=int, default=200)
    args = ap.parse_args()
    import torch  # noqa: F401
    from opensmile_amd import capi, synth
    from oracle import lldo
    ctx = capi.Context(0)
    plan = capi.Plan(ctx, capi.compare16_config())
    lens = [160000 if i % 3 else 48000 + 1600 * i for i in range(args.utts)]
    pcms = [synth.utterance(args.first + i, n) for i, n in enumerate(lens)]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    b = capi.Batch(plan, off)
    out = b.run_host(np.concatenate(pcms))
    groups = {"F0final, voicing (sma)": [0, 1], "jitter, shimmer, logHNR (sma)": [2, 3, 4, 5], "group A (sma)": list(range(6, 10)),
              "audSpec_Rfilt (sma)": list(range(10, 36)), "spectral (sma)": list(range(36, 51)), "mfcc 1-14 (sma)": list(range(51, 65)),
              "F0 group deltas": list(range(65, 71)), "groups A+B deltas": list(range(71, 130))}
    dev = {k: [] for k in groups}
    same = {k: [] for k in groups}
    rows = 0
    for i, pcm in enumerate(pcms):
        ref = lldo.run_reference_lld("compare16/ComParE_2016.conf", pcm)
        o = out[b.frame_offsets[i]:b.frame_offsets[i + 1]]
        assert o.shape == ref.shape, (i, o.shape, ref.shape)
        rows += ref.shape[0]
        scale = np.maximum(np.abs(ref[:, :65]).max(axis=0), 1e-6)
        scale = np.concatenate([scale, scale])
        rel = np.abs(o.astype(np.float64) - ref) / scale[None, :]
        eq = np.ascontiguousarray(o, np.float32).view(np.uint32) == np.ascontiguousarray(ref, np.float32).view(np.uint32)
        for k, cols in groups.items():
            dev[k].append(rel[:, cols].max(axis=1))
            same[k].append(eq[:, cols].ravel())
    res = {"utterances": args.utts, "rows": rows, "reference": "oracle/_ref/SMILExtract -C compare16/ComParE_2016.conf -lldhtkoutput"}
    for k, v in dev.items():
        d = np.concatenate(v)
        res[k] = {"max_rel": float(d.max()), "rows_over_1e-5": float((d > 1e-5).mean()), "rows_over_1e-3": float((d > 1e-3).mean()),
                  "rows_over_1e-6": float((d > 1e-6).mean()), "cells_bit_identical": float(np.concatenate(same[k]).mean())}
    eq_all = np.concatenate([np.stack(same[k][i].reshape(-1, len(groups[k])) for i in range(len(pcms))) if False else np.concatenate([x.reshape(-1, len(groups[k])) for x in same[k]]) for k in groups], axis=1)
    cols = sum((groups[k] for k in groups), [])
    res["columns_not_bit_identical"] = {str(c): float(1.0 - eq_all[:, i].mean()) for i, c in enumerate(cols) if eq_all[:, i].mean() < 1.0}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
