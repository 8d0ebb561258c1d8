#!/usr/bin/env python3
"""Runs of voiced frames in the F0 contour of the bench corpus (what lld_jitter_runs parallelises over)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from opensmile_amd import capi, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    ctx = capi.Context(0)
    plan = capi.Plan(ctx, capi.compare16_config())
    pcm, off = synth.corpus_tiled(n, 160000, n_unique=32)
    b = capi.Batch(plan, off)
    d_pcm = torch.from_numpy(pcm).cuda()
    d_out = torch.empty((b.total_rows, 130), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    b.run_device(d_pcm.data_ptr(), d_out.data_ptr(), 130, st)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    ro = np.arange(n + 1) * (b.total_rows // n)
    lens, voiced, per_utt = [], 0, []
    for u in range(min(n, 32)):
        f0 = out[ro[u]:ro[u + 1], 0]
        v = f0 > 0
        voiced += int(v.sum())
        d = np.diff(np.concatenate([[0], v.astype(np.int8), [0]]))
        s, e = np.where(d == 1)[0], np.where(d == -1)[0]
        lens.extend((e - s).tolist())
        per_utt.append((len(s), int(v.sum()), int((e - s).max()) if len(s) else 0))
    lens = np.array(lens)
    print("utterances", min(n, 32), "rows", int(ro[min(n, 32)]), "voiced rows", voiced, "runs", len(lens))
    print("run length: mean %.1f median %d max %d; >=128: %d" % (lens.mean(), np.median(lens), lens.max(), int((lens >= 128).sum())))
    print("per utterance (runs, voiced, longest):", per_utt)


if __name__ == "__main__":
    main()
