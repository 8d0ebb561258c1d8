#!/usr/bin/env python3
"""Parity statistics of the functionals level of ComParE_2016 (6373 values per utterance): GPU (whole-level chain +
smilehip_batch_functionals_compare16) against the REAL reference binary (oracle/_ref/SMILExtract -htkoutput) over fresh
synthetic utterances. Prints one JSON object: per functional (value-name suffix) the median / 99th-percentile error
relative to max(|reference|, 1e-2) and the share of values within 1e-3 and bit-identical."""
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
    _, func, _ = b.run_host_with_functionals16(np.concatenate(pcms))
    names = list(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                      "compare16_func_synth.npz"))["names"])
    suffix = np.array([str(n).rsplit("_", 1)[1] for n in names])
    ref = np.stack([lldo.run_reference_func("compare16/ComParE_2016.conf", p)[0][0] for p in pcms])
    err = np.abs(func.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-2)
    same = func.view(np.uint32) == ref.view(np.uint32)
    res = {"utterances": args.utts, "values": int(err.size), "reference": "oracle/_ref/SMILExtract -C compare16/ComParE_2016.conf -htkoutput",
           "all": {"median": float(np.median(err)), "p99": float(np.quantile(err, 0.99)), "within_1e-3": float((err <= 1e-3).mean()),
                   "bit_identical": float(same.mean())}, "by_functional": {}}
    for sfx in sorted(set(suffix), key=lambda x: -float(np.quantile(err[:, suffix == x], 0.99))):
        e = err[:, suffix == sfx]
        res["by_functional"][sfx] = {"n": int(e.size), "median": float(np.median(e)), "p99": float(np.quantile(e, 0.99)),
                                     "within_1e-3": float((e <= 1e-3).mean()), "bit_identical": float(same[:, suffix == sfx].mean())}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
